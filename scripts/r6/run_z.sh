#!/bin/bash
# round 6, batch z: the grouped K order as the planner's default — 8-bit tile tests, then the named shapes
export TMPDIR=/tmp
O=gpurun_out/r6z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -n 3 -k "a8w8 or mx or fp8 or config4 or config5 or structured or exact or fullsize or round6 or helper or processors" > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
python scripts/r6/probe_k_rotation.py fp8_16384_m256 a8w8_4096x14336_m256 a8w8_4096x8192_m256 mx_a8w8_4096x8192_m256 a8w8_4096_m256 2>&1 | grep -v "amdgpu.ids\|^Loaded" > $O/probe_k_order_defaults.log
python - <<'PY'
import json
for l in open('gpurun_out/r6z/probe_k_order_defaults.log'):
    try: d = json.loads(l)
    except Exception: continue
    f = d.get('flags')
    print(d.get('workload'), ('G%d' % (1 << ((f >> 24) - 1))) if f and f >= (1<<24) else f, d.get('chained_us'), d.get('error',''))
PY
