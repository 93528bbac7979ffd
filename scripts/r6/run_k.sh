#!/bin/bash
# round 6, batch k: groups of 32 — rows kernel against the 32-row tiles at 33 .. 64 rows; then the whole GPU suite and the bench line
export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
GL_GS=32 GL_BITS=4 GL_SHAPES="4096x4096,8192x8192,11008x4096" timeout 600 python scripts/probe_rows5.py 24 32 40 48 64 > $O/probe_g32_w4_m33_64.log 2>&1; grep "^{" $O/probe_g32_w4_m33_64.log | cut -c1-330
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err; python - <<'P'
import json
d = json.load(open('gpurun_out/r6k/bench_default.json'))
r = d['roofline']
print({k: v for k, v in list(r.items())[:14]})
for g in ('m256', 'fewrows', 'trend_m1', 'cfg5', 'cfg4', 'prefill_m2048', 'm1_bf16'):
    print(g, json.dumps(r.get(g))[:900])
P
