#!/bin/bash
# round 6, batch q2: where the x-through-LDS 8-bit rows kernel hands over to the tile kernels — other shapes, forced past the planner's budgets
export TMPDIR=/tmp
O=gpurun_out/r6q2; mkdir -p $O
GL_MS=2,3,8,16,32,48,64 timeout 600 python scripts/r6/probe_w8_rows_lds.py 4096 4096 > $O/probe_w8_rows_lds_4096_b.log 2>&1
GL_MS=2,4,8,16,24,32,64 timeout 900 python scripts/r6/probe_w8_rows_lds.py 8192 8192 > $O/probe_w8_rows_lds_8192.log 2>&1
GL_MS=2,4,8,16,32,64 timeout 900 python scripts/r6/probe_w8_rows_lds.py 14336 4096 > $O/probe_w8_rows_lds_14336x4096.log 2>&1
GL_MS=2,4,8,16,32,64 timeout 900 python scripts/r6/probe_w8_rows_lds.py 4096 14336 > $O/probe_w8_rows_lds_4096x14336.log 2>&1
cat $O/probe_w8_rows_lds_4096_b.log $O/probe_w8_rows_lds_8192.log $O/probe_w8_rows_lds_14336x4096.log $O/probe_w8_rows_lds_4096x14336.log | grep -v "^Loaded\|amdgpu.ids" | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['proc'][:12], d['N'], d['K'], d['M'], ' '.join('%s=%s' % (k[:-3], d[k]) for k in d if k.endswith('_us')), ' '.join(k for k in d if k.endswith('_err')))
"
