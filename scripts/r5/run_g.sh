#!/bin/bash
# round 5, batch g: the unsplit 128 x 128 A8W8 tile kernel: parity, then timing
mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "sq128 or a8w8_mfma_kernel_matches or a8w8_lds_kernel_short or config5" > gpurun_out/r5g/pytest.log 2>&1
tail -5 gpurun_out/r5g/pytest.log
timeout 900 python scripts/probe_a8w8_sq128.py > gpurun_out/r5g/probe_a8w8_sq128.log 2>&1
cat gpurun_out/r5g/probe_a8w8_sq128.log | cut -c1-260
