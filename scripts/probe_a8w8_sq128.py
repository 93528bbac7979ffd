"""Round 5: the unsplit 128 x 128 A8W8 tile kernel (tuning[0] = 10) against the round-3 / round-4 choices (tuning[0] = 6: both operands
through LDS with K slices, tuning[0] = 5: unsplit 64 x 64 tiles), graph-replayed time per launch over rotating layers, outputs compared
with the first candidate.  The planner's rule (a8w8_sq128_pays in api.hip) comes from this log.
    python scripts/probe_a8w8_sq128.py [workload ...]"""
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
W = bench.WORKLOADS
for m in (128, 256, 512, 1024):
    W[f"a8w8_8192_m{m}"] = (8192, 8192, 8, 8192, m, "int8", 8, "mfma")
    W[f"fp8_8192_m{m}"] = (8192, 8192, 8, 8192, m, "fp8w8", 8, "mfma")
for m in (128, 512):
    W[f"fp8_16384_m{m}"] = (16384, 16384, 8, 16384, m, "fp8w8", 2, "mfma")
W["a8w8_16384_m256"] = (16384, 16384, 8, 16384, 256, "int8", 2, "mfma")
W["a8w8_14336x4096_m256"] = (14336, 4096, 8, 4096, 256, "int8", 12, "mfma")
W["a8w8_4096x14336_m1024"] = (4096, 14336, 8, 14336, 1024, "int8", 12, "mfma")
W["a8w8_4096_m1024"] = (4096, 4096, 8, 4096, 1024, "int8", 32, "mfma")
W["a8w8_4096_m2048"] = (4096, 4096, 8, 4096, 2048, "int8", 32, "mfma")
names = ["fp8_16384_m256", "fp8_16384_m512", "a8w8_16384_m256", "a8w8_8192_m256", "a8w8_8192_m512", "a8w8_8192_m1024", "fp8_8192_m512",
         "a8w8_14336x4096_m256", "a8w8_4096x14336_m1024", "a8w8_4096_m1024", "a8w8_4096_m2048", "fp8_16384_m256"]
only = sys.argv[1:]
for name in names:
    if only and name not in only:
        continue
    first = None
    for t in ((6, 0, 0, 0), (0, 2, 8, 0), (10, 0, 4, 0), (10, 0, 5, 0), None, (10, 0, 4, 0)):
        core.TUNING_OVERRIDE = t
        try:
            r = bench.Runner(name, dev, lib)
            y = r.call(r.mods[0]).float().cpu().numpy()
            torch.cuda.synchronize()
            if first is None:
                first = y
            c_us, n, el = r.chained_us_per_launch(min_seconds=0.15)
            print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3), tops=round(r.flops / c_us / 1e6, 1),
                                  frac=round(r.flops / c_us / 1e6 / 5000, 4), equal_first=bool(np.array_equal(y, first)),
                                  rel_vs_first=float(np.abs(y - first).mean() / (np.abs(first).mean() + 1e-30)))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
