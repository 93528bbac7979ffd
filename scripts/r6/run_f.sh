#!/bin/bash
# round 6, batch f: M = 1 planner sweep after gemv_wn moved to counted asm loads (4-bit and 2-bit, 22 LLM shapes), then the M = 256 K-loop ablation
export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
timeout 900 python scripts/probe_m1_shapes.py 1 4 > $O/probe_m1_shapes_w4.log 2>&1
timeout 900 python scripts/probe_m1_shapes.py 1 2 > $O/probe_m1_shapes_w2.log 2>&1
for f in $O/probe_m1_shapes_w4.log $O/probe_m1_shapes_w2.log; do grep "^{" $f | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); a = r['all']
    wn = min((v, k) for k, v in a.items() if k.endswith('512)') and v is not None)
    mf = min(((v, k) for k, v in a.items() if k.endswith('1024)') and v is not None), default=(None, None))
    print(r['N'], r['K'], 'default', r['default'], 'best_wn', wn, 'best_mfma', mf, 'gain', r['gain'])"; done
touch gemlite_amd/csrc/gemm_wn_mma_kernel.inc
(time make -C gemlite_amd/csrc MMA_EXTRA=-DGL_MMA_EXPERIMENTS -j4) > $O/build.log 2>&1; tail -3 $O/build.log
timeout 900 python scripts/r6/ablate_m256.py > $O/ablate_m256.log 2>&1; grep "^{" $O/ablate_m256.log
