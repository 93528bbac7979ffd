"""Round 6: gemv_wn_kernel at M = 1 with 8 waves per block (tuning[2] = 8) against the planner's choice, 4-bit g128, LLM shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
SHAPES = [(8192, 8192), (5120, 5120), (6144, 4096), (8192, 4096), (4096, 8192), (8192, 3072), (12288, 4096), (11008, 4096), (14336, 4096), (4096, 14336), (5120, 13824), (13824, 5120), (8192, 28672), (28672, 8192)]
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for (N, K) in SHAPES:
    name = f"a16w{NB}_{N}x{K}_m1"
    nl = max(2, min(32, int(300e6 // (N * K * NB // 8))))
    bench.WORKLOADS[name] = (N, K, NB, 128, 1, "fp16", nl, "hbm")
    rec = dict(N=N, K=K)
    for rep in range(2):
        for vn, t in (("default", (0, 0, 0, 0)), ("w8", (0, 0, 8, 0)), ("t32w8", (3, 1, 8, 0)), ("t64w8", (4, 1, 8, 0)), ("t64sk2w8", (4, 2, 8, 0))):
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.1)
                rec[f"{vn}{rep}"] = (round(c_us, 2), r.kernel_name())
                del r
            except Exception as e:
                rec[f"{vn}{rep}"] = (None, str(e)[:40])
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
    print(json.dumps(rec), flush=True)
