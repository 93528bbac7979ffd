#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread is ~5 %): for each round, each scripts/ab/lib_<v>.so is copied over
# the in-tree library and the same probe runs in a fresh process.  Usage: bash scripts/ab_mma.sh "<variants>" <probe args>
# Build each variant first (make -C gemlite_amd/csrc at the commit / with the patch to compare) and copy
# gemlite_amd/csrc/libgemlite_hip.so to scripts/ab/lib_<name>.so (git-ignored).
V="$1"; shift
cp gemlite_amd/csrc/libgemlite_hip.so /tmp/lib_keep.so
for round in 1 2; do
  for v in $V; do
    cp scripts/ab/lib_$v.so gemlite_amd/csrc/libgemlite_hip.so
    echo "== $v (round $round)"
    python scripts/probe_ab.py "$@" 2>&1 | grep '^{'
  done
done
cp /tmp/lib_keep.so gemlite_amd/csrc/libgemlite_hip.so
