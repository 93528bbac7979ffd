#!/bin/bash
# round 5, GPU call F: x pieces requested BEFORE the weights at a piece end (in-order return per wave) — long-K shapes; pack kernel test
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "rows5 or pack_unpack" -p no:cacheprovider --timeout 600 > $O/pytest_rows5.log 2>&1; echo "rc=$?" >> $O/pytest_rows5.log; tail -6 $O/pytest_rows5.log
timeout 900 python -m pytest tests/test_structured_exact_gpu.py -q -k "packed_weight_families" -p no:cacheprovider --timeout 600 > $O/pytest_struct.log 2>&1; echo "rc=$?" >> $O/pytest_struct.log; tail -4 $O/pytest_struct.log
GL_SHAPES=4096x4096,4096x8192,4096x11008,4096x14336,3072x8192,4096x2048 timeout 900 python scripts/probe_rows5.py 2 8 16 32 64 > $O/probe_rows5_ns4.log 2>&1
grep -v "^/opt\|^Loaded" $O/probe_rows5_ns4.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['N'],d['K'],d['M'],d['us'],d['r4_kernel'],d['x_reread_MB'])
"
python - <<'PY'
import torch, time
from gemlite_amd import bitpack
W = torch.randint(0, 16, (4096, 4096), dtype=torch.int32).to(torch.uint8).cuda()
for _ in range(3): bitpack.pack_weights_over_cols(W, 4, 32, True)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(50): bitpack.pack_weights_over_cols(W, 4, 32, True)
torch.cuda.synchronize(); print("pack 4096^2 4-bit: %.1f us per call (incl. host)" % ((time.perf_counter()-t0)/50*1e6))
PY
