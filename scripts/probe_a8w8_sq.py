"""Round 4: the unsplit 64 x 64 A8W8 tile kernel (tuning[0] = 5) against the round-3 choice (tuning[0] = 6), graph-replayed time per
launch, outputs compared bitwise (int8 accumulates exactly; fp8 sums in a different order: tolerance).
    python scripts/probe_a8w8_sq.py"""
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
for m in (65, 96, 128, 192, 320, 384, 512, 1024):
    W[f"a8w8_4096_m{m}"] = (4096, 4096, 8, 4096, m, "int8", 32, "mfma")
for m in (128, 256, 512):
    W[f"a8w8_8192_m{m}"] = (8192, 8192, 8, 8192, m, "int8", 8, "mfma")
W["a8w8_11008x4096_m256"] = (11008, 4096, 8, 4096, 256, "int8", 12, "mfma")
W["a8w8_4096x11008_m256"] = (4096, 11008, 8, 11008, 256, "int8", 12, "mfma")
W["fp8_4096_m256"] = (4096, 4096, 8, 4096, 256, "fp8w8", 32, "mfma")
W["fp8_8192_m256"] = (8192, 8192, 8, 8192, 256, "fp8w8", 8, "mfma")
names = ["a8w8_4096_m256", "a8w8_4096_m65", "a8w8_4096_m96", "a8w8_4096_m128", "a8w8_4096_m192", "a8w8_4096_m320", "a8w8_4096_m384", "a8w8_4096_m512", "a8w8_4096_m1024",
         "a8w8_8192_m128", "a8w8_8192_m256", "a8w8_8192_m512", "a8w8_11008x4096_m256", "fp8_4096_m256", "fp8_8192_m256", "fp8_16384_m256", "a8w8_4096_m256"]
only = sys.argv[1:]
for name in names:
    if only and name not in only:
        continue
    first = None
    for t in ((6, 0, 0, 0), (5, 0, 0, 0), (5, 0, 2, 0), (5, 0, 3, 0), (5, 0, 0, 0), (5, 0, 2, 0)):
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
