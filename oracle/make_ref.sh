#!/bin/bash
# TEST INFRASTRUCTURE — stage the reference package (Python + Triton, read-only at /root/reference) under oracle/_ref/ so
# that it travels to the GPU box with `gpurun` (oracle/_ref/ is git-ignored: reference SOURCES never enter this repo's
# history).  Only oracle/run_ref_gpu.py imports it, to (a) dump reference outputs at BASELINE sizes into golden fixtures and
# (b) time the reference's Triton kernels on the same MI355X.  Run in the build container:  bash oracle/make_ref.sh
set -e
cd "$(dirname "$0")"
SRC=${GEMLITE_REFERENCE:-/root/reference}
[ -d "$SRC/gemlite" ] || { echo "no reference at $SRC (GPU box?): nothing staged"; exit 0; }
rm -rf _ref/gemlite
mkdir -p _ref
cp -r "$SRC/gemlite" _ref/gemlite
find _ref -name __pycache__ -type d -prune -exec rm -rf {} +
echo "staged $(find _ref/gemlite -name '*.py' | wc -l) reference files under oracle/_ref/gemlite"
