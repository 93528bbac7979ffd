"""Round 4: the narrow-tile rule of plan_gemm_wn_mma against the round-3 choice on the LLM layer shapes of the planner golden, at M = 128 /
192 / 256 (bf16, 4-bit g128): R3 = tuning[3] & 16384, auto = the library's rule, and every forced narrow candidate.
    python scripts/probe_mma_narrow_shapes.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
SHAPES = [(1024, 4096), (1536, 8960), (2048, 8192), (2560, 9728), (3072, 8192), (4096, 1024), (4096, 4096), (4096, 11008), (4096, 14336), (5120, 5120),
          (5120, 13824), (6144, 4096), (8192, 2048), (8192, 3072), (8192, 8192), (8960, 1536), (11008, 4096), (12288, 4096), (13824, 5120), (14336, 4096)]
Ms = [int(a) for a in sys.argv[1:]] or [256, 128, 192]
for M in Ms:
    for (N, K) in SHAPES:
        name = f"a16w4_{N}x{K}_m{M}"
        layers = max(2, min(64, (560 << 20) // (N * K // 2)))
        bench.WORKLOADS[name] = (N, K, 4, 128, M, "bf16", layers, "mfma")
        res = {}
        cands = [("r3", (0, 0, 0, 16384)), ("auto", (0, 0, 0, 0)), ("n64", (0, 1, 32, 0)), ("n64x2", (0, 2, 32, 0)), ("n128", (0, 1, 34, 0)), ("n128x2", (0, 2, 34, 0)), ("r3b", (0, 0, 0, 16384))]
        kn = {}
        for label, t in cands:
            if K % 256 != 0 and label.startswith("n"):
                continue
            core.TUNING_OVERRIDE = t
            try:
                r = bench.Runner(name, dev, lib)
                r.chained_us_per_launch(min_seconds=0.03)
                c_us, _, _ = r.chained_us_per_launch(min_seconds=0.12)
                res[label] = round(c_us, 2)
                kn[label] = r.kernel_name()
                del r
            except Exception as ex:
                res[label] = None
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
        base = min(v for k, v in res.items() if k in ("r3", "r3b") and v)
        best = min((v, k) for k, v in res.items() if v and k not in ("auto",))
        print(json.dumps(dict(M=M, N=N, K=K, t64=(N // 64) * ((M + 63) // 64), x_MB=round((N // 64) * ((M + 63) // 64 * 64) * K * 2 / 2**20), us=res, auto_kernel=kn.get("auto"),
                              r3_kernel=kn.get("r3"), auto_vs_r3=round(res["auto"] / base, 3) if res.get("auto") else None, best=best[1], best_vs_r3=round(best[0] / base, 3))), flush=True)
