#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c19; mkdir -p $O
timeout 1200 python -m pytest tests/test_mx_gpu.py tests/test_reference_suite_gpu.py tests/test_structured_exact_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider --timeout 600 -k "NVFP or nvfp or odd_shapes or acceptance or reference or mx or MX" > $O/pytest_nv.log 2>&1; tail -15 $O/pytest_nv.log | cut -c1-400
timeout 600 python scripts/probe_nvfp4.py > $O/probe_nvfp4.log 2>&1; grep -v amdgpu.ids $O/probe_nvfp4.log | cut -c1-300
