#!/bin/bash
# round 6, batch b: gemv_mfma on counted asm loads + exact planes: timing against the round-5 numbers, then the GPU suite
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "small_magnitude" -p no:cacheprovider > $O/pytest_small.log 2>&1; tail -3 $O/pytest_small.log
timeout 900 python scripts/probe_gemv3.py a16w4_8192_m1 a16w2_16384_m1 a16w4_16384_m1 a16w2_8192_m1 a16w2_4096_m1 a16w4_11008n_m1 '--tunings=[[0,0,0,0],[0,0,0,512],[0,0,0,1024],[22,0,0,1024],[24,0,0,1024],[22,0,4,1024],[24,0,4,1024]]' > $O/probe_gemv_r6.log 2>&1
grep "^{" $O/probe_gemv_r6.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('workload'), r.get('tuning'), r.get('kernel'), r.get('chained_us'), r.get('frac'), '%.2e' % r.get('rel_vs_first', -1) if 'rel_vs_first' in r else r.get('error'))"
timeout 600 python scripts/probe_gemv3.py a16w4_4096_m2 a16w4_4096_m4 a16w4_8192_m4 '--tunings=[[0,0,0,0],[0,0,0,512],[0,0,0,65536]]' > $O/probe_rows4_r6.log 2>&1
grep "^{" $O/probe_rows4_r6.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('workload'), r.get('tuning'), r.get('kernel'), r.get('chained_us'), r.get('frac'), '%.2e' % r.get('rel_vs_first', -1) if 'rel_vs_first' in r else r.get('error'))"
timeout 1800 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
