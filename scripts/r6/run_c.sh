#!/bin/bash
# round 6, batch c: what each part of the M = 256 K loops costs (ablation build on the box; the shipped library is not touched)
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
touch gemlite_amd/csrc/gemm_wn_mma_kernel.inc
(time make -C gemlite_amd/csrc MMA_EXTRA=-DGL_MMA_EXPERIMENTS -j4) > $O/build.log 2>&1; tail -3 $O/build.log
timeout 900 python scripts/r6/ablate_m256.py > $O/ablate_m256.log 2>&1; grep "^{" $O/ablate_m256.log
