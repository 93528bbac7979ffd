#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c21; mkdir -p $O
timeout 1200 python -m pytest tests/test_mx_gpu.py tests/test_reference_suite_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider --timeout 600 > $O/pytest_mx.log 2>&1; tail -15 $O/pytest_mx.log | cut -c1-600
timeout 700 python scripts/probe_mx_rows.py > $O/probe_mx_rows.log 2>&1; grep -v amdgpu.ids $O/probe_mx_rows.log | cut -c1-300
