#!/bin/bash
# round 5, GPU call C: rows kernel v2 (x through LDS-DMA in whole cache lines): parity, structured exactness, timing against round 4
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "rows5" -p no:cacheprovider --timeout 600 > $O/pytest_rows5.log 2>&1; echo "rc=$?" >> $O/pytest_rows5.log; tail -8 $O/pytest_rows5.log
timeout 900 python -m pytest tests/test_structured_exact_gpu.py -q -k "packed_weight_families" -p no:cacheprovider --timeout 600 > $O/pytest_struct.log 2>&1; echo "rc=$?" >> $O/pytest_struct.log; tail -5 $O/pytest_struct.log
timeout 600 python scripts/probe_rows5.py 2 4 8 16 24 32 48 64 > $O/probe_rows5.log 2>&1; grep -v "^/opt\|^Loaded" $O/probe_rows5.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['N'],d['K'],d['M'],d['us'],d['r4_kernel'],d['x_reread_MB'])
"
