"""Round 4: a8w8_decode_kernel (one wave per column, the weight row in flight before anything else; default) against the round-2 kernels
(tuning[0] = 7: kmajor_matmul_kernel / kmajor_fused_quant_kernel) at M = 1: matmul on pre-quantised x and layer(x) (fused quantiser).
    python scripts/probe_a8w8_decode.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
W = bench.WORKLOADS
names = []
for N, K, nl in ((4096, 4096, 32), (8192, 8192, 8), (4096, 14336, 10), (14336, 4096, 10), (8192, 28672 // 4 * 4, 4), (16384, 16384, 2), (2048, 8192, 32), (1024, 4096, 64)):
    if K % 1024:
        continue
    for dt in ("int8", "fp8w8"):
        n = f"a8_{dt}_{N}x{K}_m1"
        W[n] = (N, K, 8, K, 1, dt, nl, "hbm")
        names.append(n)
for name in names:
    for e2e in (False, True):
        rec = dict(workload=name, e2e=e2e)
        ys = []
        for label, t in (("new", None), ("old", (7, 0, 0, 0))):
            core.TUNING_OVERRIDE = t
            try:
                if e2e and t is not None:
                    # layer(x) consults TUNING_OVERRIDE for the fused decision: the fused call needs tuning through _hip_matmul
                    pass
                r = bench.Runner(name, dev, lib, e2e=e2e)
                y = r.call(r.mods[0]).float()
                torch.cuda.synchronize()
                ys.append(y)
                us, n, el = r.chained_us_per_launch(min_seconds=0.12)
                rec[label + "_us"] = round(us, 3)
                rec[label + "_kernel"] = r.kernel_name()
                del r
            except Exception as e:
                rec[label + "_us"] = f"{type(e).__name__}: {e}"[:120]
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
        if len(ys) == 2:
            rec["rel"] = float((ys[0] - ys[1]).abs().mean() / (ys[1].abs().mean() + 1e-30))
        print(json.dumps(rec), flush=True)
