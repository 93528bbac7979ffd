#!/bin/bash
# round 6, batch q: the x-through-LDS 8-bit rows kernel — parity tests, then the A/B against the round-4 kernels
export TMPDIR=/tmp
O=gpurun_out/r6q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -k "rows_lds or a16w8_rows or a8w8_rows or a8w8_families or test_fp8_e5m2 or a16w8_tile" > $O/pytest_sub.log 2>&1; tail -25 $O/pytest_sub.log
timeout 900 python scripts/r6/probe_w8_rows_lds.py > $O/probe_w8_rows_lds_4096.log 2>&1; grep -v "^Loaded\|amdgpu.ids" $O/probe_w8_rows_lds_4096.log
