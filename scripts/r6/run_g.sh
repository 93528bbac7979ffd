#!/bin/bash
# round 6, batch g: the M = 1 planner after the rule change (default against the candidates again), then the bench line
export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python scripts/probe_m1_shapes.py 1 4 > $O/probe_m1_shapes_w4_after.log 2>&1
timeout 900 python scripts/probe_m1_shapes.py 1 2 > $O/probe_m1_shapes_w2_after.log 2>&1
for f in $O/probe_m1_shapes_w4_after.log $O/probe_m1_shapes_w2_after.log; do grep "^{" $f | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['N'], r['K'], 'default', r['default'], 'best', r['best'], 'gain', r['gain'])"; done
timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err; python - <<'P'
import json
d = json.load(open('gpurun_out/r6g/bench_default.json'))
r = d['roofline']
print({k: v for k, v in list(r.items())[:14]})
for g in ('m256', 'fewrows', 'trend_m1', 'cfg5', 'm1_bf16'):
    print(g, json.dumps(r.get(g)))
P
