#!/bin/bash
# round 6, batch j: groups of 32 on the tile kernel: parity, then what it buys against the coverage kernel / the rows kernel
export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "groups_of_32" -p no:cacheprovider > $O/pytest_g32.log 2>&1; tail -12 $O/pytest_g32.log
GL_GS=32 GL_BITS=1 GL_SHAPES="4096x4096" timeout 600 python scripts/probe_rows5.py 8 64 256 > $O/probe_g32_w1.log 2>&1; grep "^{" $O/probe_g32_w1.log | cut -c1-330
GL_GS=32 GL_BITS=4 GL_SHAPES="4096x4096" timeout 600 python scripts/probe_rows5.py 128 256 1024 > $O/probe_g32_w4.log 2>&1; grep "^{" $O/probe_g32_w4.log | cut -c1-330
