#!/bin/bash
# round 6, batch e: gemv_wn_kernel on counted asm loads (gvw::ring2_run): parity, then timing against batch d (same shapes)
export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py -q -m gpu -k "small_magnitude or gemv or decode or structured or cfg or bitwidth or all_weight_modes or weight_modes" -p no:cacheprovider -x > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
timeout 900 python scripts/probe_gemv3.py a16w4_8192_m1 a16w2_16384_m1 a16w4_16384_m1 a16w2_8192_m1 a16w2_4096_m1 a16w4_11008n_m1 a16w4_11008_m1 a16w4_4096_m1 '--tunings=[[0,0,0,0],[0,0,0,512],[0,0,0,1024],[3,0,0,512],[4,0,0,512],[4,0,8,512]]' > $O/probe_gemv_r6.log 2>&1
grep "^{" $O/probe_gemv_r6.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('workload'), r.get('tuning'), r.get('kernel'), r.get('chained_us'), r.get('frac'), '%.2e' % r.get('rel_vs_first', -1) if 'rel_vs_first' in r else r.get('error'))"
