#!/bin/bash
# round 5, batch h: rows kernel with two column tiles per block (4096 < N <= 8192): parity, then timing; the tests whose name asserts changed
mkdir -p gpurun_out/r5h2
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py -q -x -m gpu -k "rows5 or direct_mfma" > gpurun_out/r5h2/pytest.log 2>&1
tail -5 gpurun_out/r5h2/pytest.log
GL_SHAPES="8192x8192,8192x4096,6144x4096,5120x5120,8192x2048,4096x4096" timeout 900 python scripts/probe_rows5.py 4 8 16 32 64 > gpurun_out/r5h2/probe_rows5_nt2.log 2>&1
grep "^{" gpurun_out/r5h2/probe_rows5_nt2.log | cut -c1-250
GL_SHAPES="8192x8192,6144x4096" GL_GS=64 GL_DT=bf16 timeout 600 python scripts/probe_rows5.py 8 16 32 > gpurun_out/r5h2/probe_rows5_nt2_g64_bf16.log 2>&1
grep "^{" gpurun_out/r5h2/probe_rows5_nt2_g64_bf16.log | cut -c1-250
