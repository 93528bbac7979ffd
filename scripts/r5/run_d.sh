#!/bin/bash
# round 5, GPU call D: rows kernel on narrower / shorter layers (planner rule), then the whole GPU suite on the new defaults
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
GL_SHAPES=3072x8192,4096x1024,4096x2048,4096x8192,2048x8192,3584x4096 timeout 600 python scripts/probe_rows5.py 2 8 16 32 64 > $O/probe_rows5_more.log 2>&1
grep -v "^/opt\|^Loaded" $O/probe_rows5_more.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['N'],d['K'],d['M'],d['us'],d['r4_kernel'],d['default_kernel'],d['x_reread_MB'])
"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -n 4 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
