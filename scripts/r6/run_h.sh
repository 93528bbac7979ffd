#!/bin/bash
# round 6, batch h: packed words through LDS in the M = 256 tile kernels: A/B + bit-identity, then the tile-kernel parity tests
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
timeout 900 python scripts/r6/probe_mma_wl.py > $O/probe_mma_wl.log 2>&1; grep "^{" $O/probe_mma_wl.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py tests/test_ref_fullsize_gpu.py -q -m gpu -k "tiled or mma or m256 or cfgA or cfgB or structured or reference_outputs or narrow or prefill or bitwidth" -p no:cacheprovider > $O/pytest_sub.log 2>&1; tail -8 $O/pytest_sub.log
