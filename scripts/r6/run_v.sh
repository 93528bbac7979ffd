#!/bin/bash
# round 6, batch v: the reference pins of the second half of the round + the tests around the GEMV change (group index with gs_magic)
export TMPDIR=/tmp
O=gpurun_out/r6v; mkdir -p $O
timeout 1500 python -m pytest tests/test_ref_fullsize_gpu.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 900 -n 3 -k "fullsize or round6 or power_of_two or gemv or decode or cfgA or m1 or rows_lds" > $O/pytest_sub.log 2>&1; tail -8 $O/pytest_sub.log
python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0, '.')
import gemlite_amd
from oracle.run_ref_gpu import CASES
import tests.test_gpu_parity as T
for name, build, _ in CASES:
    if name in ("w4_g96_fp16_m1", "w4_g96_fp16_m16"):
        lin, x = build(gemlite_amd)
        print(name, T._kernel_name(lin, x))
PY
for w in a16w4_4096_m1 a16w4_8192_m1 a16w2_16384_m1; do timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:30], d['roofline']['kernel'], d['roofline']['kernel_us'])"; done
