"""A8W8 int8 / fp8 at 17..64 rows: the 16-column rows kernel with 2 / 4 row tiles (default from round 3) against the 32- / 64-row
tiles of the 8-wave MFMA kernel (tuning[2] = 1 / 2), graph-replayed time per launch, HBM-cold rotating layers.
    python scripts/probe_a8w8_rows.py"""
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
SHAPES = {"4096": (4096, 4096, 16), "8192": (8192, 8192, 6), "14336x4096": (14336, 4096, 6), "4096x14336": (4096, 14336, 6)}
for dt in ("int8", "fp8"):
    for sname, (N, K, nl) in SHAPES.items():
        if dt == "fp8" and sname not in ("4096", "8192"):
            continue
        for M in (17, 24, 32, 48, 64):
            name = f"a8w8_{dt}_{sname}_m{M}"
            bench.WORKLOADS[name] = (N, K, 8, K, M, dt, nl, "hbm")
            first = None
            for t in ((0, 0, 0, 0), (0, 0, 1, 0), (0, 0, 2, 0)):
                if t[2] == 2 and M <= 32:
                    continue
                core.TUNING_OVERRIDE = t if any(t) else None
                try:
                    r = bench.Runner(name, dev, lib)
                    y = r.call(r.mods[0]).float().cpu().numpy()
                    torch.cuda.synchronize()
                    if first is None:
                        first = y
                    c_us, n, el = r.chained_us_per_launch(min_seconds=0.15)
                    print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3), gbs=round(r.bytes / 1e3 / c_us, 1),
                                          equal_first=bool(np.array_equal(y, first)), rel=float(np.abs(y - first).mean() / np.abs(first).mean()))), flush=True)
                    del r
                except Exception as e:
                    print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
                finally:
                    core.TUNING_OVERRIDE = None
                torch.cuda.empty_cache()
