"""Round 6: gemm_a8w8_sq_kernel<64x64> (BASELINE config 4 at M = 256) by number of LDS stages — is the loop bound by bytes in flight?
    python scripts/r6/probe_a8w8_sq_stages.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for name in ("a8w8_4096_m256",):
    first = None
    for rep in range(2):
        for t in ((0, 0, 0, 0), (5, 0, 2, 0), (5, 0, 3, 0), (5, 0, 4, 0), (5, 0, 5, 0)) + tuple((5, 0, 0, f) for f in (int(v) for v in os.environ.get("GL_FLAGS", "").split(",") if v)):
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                y = r.call(r.mods[0]).float().cpu().numpy()
                torch.cuda.synchronize()
                if first is None:
                    first = y
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.3)
                print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3), equal_first=bool(np.array_equal(y, first)))), flush=True)
                del r
            except Exception as e:
                print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
