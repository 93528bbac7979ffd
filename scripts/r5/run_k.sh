#!/bin/bash
# round 5, batch k: the rows kernel on 2-bit words: parity, then timing against the round-4 choice on LLM shapes
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py -q -x -m gpu -k "rows5" > gpurun_out/r5k/pytest.log 2>&1
tail -4 gpurun_out/r5k/pytest.log
GL_BITS=2 GL_SHAPES="4096x4096,8192x8192,4096x14336,11008x4096,2048x8192,6144x4096,16384x16384" timeout 900 python scripts/probe_rows5.py 2 4 8 16 32 48 64 > gpurun_out/r5k/probe_rows5_w2.log 2>&1
grep "^{" gpurun_out/r5k/probe_rows5_w2.log | cut -c1-300
GL_BITS=2 GL_GS=64 GL_DT=bf16 GL_SHAPES="4096x4096,8192x8192" timeout 600 python scripts/probe_rows5.py 8 16 32 64 > gpurun_out/r5k/probe_rows5_w2_g64_bf16.log 2>&1
grep "^{" gpurun_out/r5k/probe_rows5_w2_g64_bf16.log | cut -c1-300
