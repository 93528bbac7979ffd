#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c16; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --timeout 600 -s > $O/pytest_gpu.log 2>&1; grep "small-x" $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log | cut -c1-300
