#!/bin/bash
# round 6, batch a: reference outputs for the rows kernel's territory + the bench line with the new key order
export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python oracle/run_ref_gpu.py --which ref --only r5 --fixture fullsize_ref_r5.npz --budget-s 700 --out $O > $O/ref.log 2>&1; echo "ref rc=$?"
timeout 600 python oracle/run_ref_gpu.py --which hip --only r5 --out $O > $O/hip.log 2>&1; echo "hip rc=$?"
grep -h "^{" $O/hip.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('case'), r.get('rel_mean_hip_vs_ref'), r.get('rel_max_hip_vs_ref'), r.get('graph_us'), r.get('error'))"
grep -h "^{" $O/ref.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('REF', r.get('case'), r.get('first_call_s'), r.get('graph_us'), r.get('error'))"
timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err; python - <<'P'
import json
d = json.load(open('gpurun_out/r6a/bench_default.json'))
r = d['roofline']
print({k: v for k, v in list(r.items())[:14]})
for g in ('m256', 'fewrows', 'trend_m1', 'cfg5', 'cfg4', 'prefill_m2048'):
    print(g, json.dumps(r.get(g)))
P
tail -3 $O/bench_default.err
