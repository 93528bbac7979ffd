#!/bin/bash
# round 5, GPU call E: rows kernel v3 (persistent grid, weights 4 chunks ahead): parity, structured exactness, timing on 14 shapes
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "rows5" -p no:cacheprovider --timeout 600 -x > $O/pytest_rows5.log 2>&1; echo "rc=$?" >> $O/pytest_rows5.log; tail -8 $O/pytest_rows5.log
timeout 900 python -m pytest tests/test_structured_exact_gpu.py -q -k "packed_weight_families or rows5" -p no:cacheprovider --timeout 600 > $O/pytest_struct.log 2>&1; echo "rc=$?" >> $O/pytest_struct.log; tail -8 $O/pytest_struct.log
GL_SHAPES=4096x4096,8192x8192,4096x11008,11008x4096,4096x14336,14336x4096,6144x4096,5120x5120,4096x8192 timeout 900 python scripts/probe_rows5.py 2 8 16 32 64 > $O/probe_rows5_v3.log 2>&1
grep -v "^/opt\|^Loaded" $O/probe_rows5_v3.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['N'],d['K'],d['M'],d['us'],d['r4_kernel'],d['x_reread_MB'])
"
