"""Few rows (M = 8, 16) of A16W4 g128 on LLM shapes: the planner's choice against (tile width, K slices, waves) candidates of
gemm_wn_direct.hip — graph-replayed us per launch, HBM-cold rotating layers.    python scripts/probe_fewrows.py"""
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
SHAPES = [(4096, 4096), (8192, 8192), (11008, 4096), (14336, 4096), (4096, 14336), (6144, 4096), (5120, 5120), (13824, 5120), (28672, 8192), (8192, 28672)]
CANDS = [(0, 0, 0, 0), (1, 1, 4, 0), (2, 1, 4, 0), (2, 1, 8, 0), (2, 2, 4, 0), (2, 2, 8, 0), (4, 1, 4, 0), (4, 1, 8, 0), (4, 2, 4, 0), (4, 2, 8, 0), (4, 4, 8, 0)]
only_m = [int(a) for a in sys.argv[1:]] or [8, 16]
if max(only_m) > 16:   # 17 .. 32 rows: two row tiles (4 waves only) against the 32-row tile of the 8-wave MFMA kernel (tuning[0] = 3)
    CANDS = [(0, 0, 0, 0), (2, 1, 0, 0), (2, 2, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (4, 4, 0, 0), (3, 0, 0, 0), (3, 2, 0, 0), (3, 4, 0, 0)]
for (N, K) in SHAPES:
    for M in only_m:
        name = f"a16w4_{N}x{K}_m{M}"
        nl = max(2, min(32, int(300e6 // (N * K // 2))))
        bench.WORKLOADS[name] = (N, K, 4, 128, M, "fp16", nl, "hbm")
        res = {}
        for t in CANDS:
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                kn = r.kernel_name()
                if any(t) and t[0] != 3 and "direct" not in kn:
                    raise RuntimeError("not the direct kernel: " + kn)
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.1)
                res[str(t)] = (round(c_us, 2), kn)
                del r
            except Exception as e:
                res[str(t)] = (None, str(e)[:60])
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
        ok = {k: v for k, v in res.items() if v[0] is not None}
        best = min(ok, key=lambda k: ok[k][0])
        print(json.dumps(dict(N=N, K=K, M=M, default=res["(0, 0, 0, 0)"], best=[best, ok[best][0]], all={k: v[0] for k, v in res.items()})), flush=True)
