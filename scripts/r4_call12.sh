#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "a8w8 or cfgA or forward_functional or bias or hip_graph or determinism or autotune or shipped or narrow or unsupported" > $O/pytest_sel.log 2>&1; tail -6 $O/pytest_sel.log
python - <<'PY' 2>&1 | grep -v amdgpu
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
from gemlite_amd import _hip, core
lib = _hip.load()
dev = torch.device("cuda:0")
print("fast path module:", core._FAST)
for name in ("a16w4_4096_m1", "a16w4_4096_m16", "a16w4_4096_m256"):
    r = bench.Runner(name, dev, lib, layers=8, use_graph=False)
    lin, x = r.mods[0], r.x
    y_fast = lin(x); y_fast2 = lin(x)
    d = lin.__dict__.get("_fast")
    os.environ["X"] = "1"
    saved = core._FAST
    # slow path result for comparison
    lin.__dict__["_fast"] = None; lin.__dict__["_fast_tried"] = core._CACHE_EPOCH[0]
    y_slow = lin(x)
    torch.cuda.synchronize()
    print(name, "handle installed:", d is not None, "fast == slow:", bool(torch.equal(y_fast2, y_slow)))
    lin.__dict__["_fast_tried"] = None
    for mode in ("fast", "slow"):
        for m in r.mods:
            m.__dict__["_fast"] = None
            m.__dict__["_fast_tried"] = None if mode == "fast" else core._CACHE_EPOCH[0]
        h, e = r.eager_us_per_call(calls=4000)
        print(f"  {mode}: host {h:.2f} us per call, incl. device {e:.2f}")
PY
