#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c11; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --timeout 600 > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
timeout 300 python scripts/timeline_mma3.py cfgA_r4 > $O/timeline_cfgA_narrow.log 2>&1; grep -v amdgpu $O/timeline_cfgA_narrow.log | cut -c1-900
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json
d=json.load(open('$O/bench_default.json'))
print('value', d['value'], d['roofline']['kernel'], d['roofline']['kernel_us'], d['roofline']['frac'])
for k in ('roofline_m256','roofline_cfg4','roofline_cfg5','roofline_trend_m1'):
    for kk,v in d[k].items(): print(k, kk, v['kernel'], v['kernel_us'], v['frac'])
print('bf16', d['roofline_m1_bf16']['kernel_us'], 'prefill', d['roofline_prefill_m2048']['kernel_us'], d['roofline_prefill_m2048']['frac'], 'eager', d['eager'])
"; tail -3 $O/bench_default.err
