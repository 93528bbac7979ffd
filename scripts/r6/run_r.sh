#!/bin/bash
# round 6, batch r: the whole GPU suite after the planner moved the 8-bit few-row families to gemm_w8_rows.hip, then the bench line
export TMPDIR=/tmp
O=gpurun_out/r6x; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r6x/bench_default.json'))
r = d['roofline']
print(d['value'], r['kernel_us'], r['m256_cfgA_us'], r['m256_cfgB_us'])
for k, b in list(r['cfg4'].items()) + list(r['mx_m256'].items()) + list(r['cfg5'].items()):
    print(k, b)
PY
tail -3 $O/bench_default.err
