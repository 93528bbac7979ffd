"""A8W8 int8 / fp8 at M = 256 (BASELINE config 4): tile height x K slices, graph-replayed time per launch.
    python scripts/probe_a8w8_m256.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
bench.WORKLOADS.update({"a8w8_4096_m64": (4096, 4096, 8, 4096, 64, "int8", 16, "mfma"), "a8w8_4096_m128": (4096, 4096, 8, 4096, 128, "int8", 16, "mfma"),
                        "a8w8_4096_m32": (4096, 4096, 8, 4096, 32, "int8", 16, "mfma")})
for name, tunings in (("a8w8_4096_m256", [(0, 0, 0, 0), (0, 1, 1, 0), (0, 2, 1, 0), (0, 1, 2, 0), (0, 2, 2, 0), (0, 2, 4, 0), (0, 4, 4, 0), (0, 1, 4, 0), (0, 4, 8, 0), (0, 2, 4, 64)]),
                      ("a8w8_4096_m128", [(0, 0, 0, 0), (0, 1, 1, 0), (0, 2, 1, 0), (0, 2, 2, 0), (0, 4, 4, 0)]),
                      ("a8w8_4096_m64", [(0, 0, 0, 0), (0, 1, 1, 0), (0, 2, 1, 0), (0, 4, 1, 0), (0, 4, 2, 0)]),
                      ("a8w8_4096_m32", [(0, 0, 0, 0), (0, 2, 1, 0), (0, 4, 1, 0)])):
    first = None
    for t in tunings:
        core.TUNING_OVERRIDE = t
        try:
            r = bench.Runner(name, dev, lib)
            y = r.call(r.mods[0]).float().cpu().numpy()
            torch.cuda.synchronize()
            if first is None:
                first = y
            c_us, n, el = r.chained_us_per_launch(min_seconds=0.2)
            print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3), tops=round(r.flops / c_us / 1e6, 1),
                                  frac=round(r.flops / c_us / 1e6 / 5000, 4), equal_first=bool(np.array_equal(y, first)))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
