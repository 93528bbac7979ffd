#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests/test_reference_suite_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_refsuite.log 2>&1; tail -25 $O/pytest_refsuite.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
def show(k,v,ind=0):
    if isinstance(v,dict) and 'kernel_us' in v: print(' '*ind+k, v.get('kernel'), v['kernel_us'], v.get('frac'), v.get('what','')[:50])
    elif isinstance(v,dict):
        print(' '*ind+k+':')
        for kk,vv in v.items(): show(kk,vv,ind+2)
    else: print(' '*ind+k, str(v)[:120])
for k,v in d.items(): show(k,v)
"; tail -3 $O/bench_default.err
