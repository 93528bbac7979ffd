"""Round 6: cfgB (A16W4 g128 8192^2 M = 256 bf16) on other (tile height, K slices, combine) forms of the tile kernel than the planner's 128 x 128 x 2 slices."""
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
name = sys.argv[1] if len(sys.argv) > 1 else "a16w4_8192_m256"
first = None
for rep in range(2):
    for t in ((0, 0, 0, 0), (0, 2, 4, 0), (0, 2, 4, 2048), (0, 2, 4, 128), (0, 4, 8, 0), (0, 4, 8, 2048), (0, 2, 8, 0), (0, 4, 4, 0), (0, 3, 4, 0), (0, 1, 4, 0), (0, 4, 2, 0), (0, 8, 8, 0)):
        core.TUNING_OVERRIDE = t if any(t) else None
        try:
            r = bench.Runner(name, dev, lib)
            y = r.call(r.mods[0]).float().cpu().numpy()
            torch.cuda.synchronize()
            if first is None:
                first = y
            c_us, n, el = r.chained_us_per_launch(min_seconds=0.3)
            print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3), rel=float(np.abs(y - first).mean() / np.abs(first).mean()))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:160])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
