"""Round 6: K rotation between the row tiles that share a weight column tile (gemm_wn_mma_kernel.inc / gemm_a8w8_sq_kernel) — default (rotation where the
siblings share an XCD) | tuning[3] & 4194304 (none) | & 8388608 (every tile form), one box, one process, graph-replayed us over rotating HBM-cold layers.
    python scripts/r6/probe_k_rotation.py [workload ...]"""
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
names = sys.argv[1:] or ["a16w4_4096_m256", "a16w4_8192_m256", "a16w4_8192_m2048", "a16w4_4096_m2048", "a8w8_4096_m256", "a16w2_16384_m256"]
bench.WORKLOADS.setdefault("a16w4_4096_m192", (4096, 4096, 4, 128, 192, "bf16", 32, "mfma"))
bench.WORKLOADS.setdefault("a16w4_4096_m128", (4096, 4096, 4, 128, 128, "bf16", 32, "mfma"))
bench.WORKLOADS.setdefault("a16w4_11008_m256", (11008, 4096, 4, 128, 256, "bf16", 16, "mfma"))
bench.WORKLOADS.setdefault("a16w4_8192x4096_m256", (8192, 4096, 4, 128, 256, "bf16", 16, "mfma"))
bench.WORKLOADS.setdefault("a8w8_16384_m256", (16384, 16384, 8, 16384, 256, "int8", 2, "mfma"))
bench.WORKLOADS.setdefault("fp8_8192_m512", (8192, 8192, 8, 8192, 512, "fp8w8", 8, "mfma"))
bench.WORKLOADS.setdefault("a8w8_8192_m256", (8192, 8192, 8, 8192, 256, "int8", 8, "mfma"))
bench.WORKLOADS.setdefault("fp8_8192_m256", (8192, 8192, 8, 8192, 256, "fp8w8", 8, "mfma"))
bench.WORKLOADS.setdefault("a8w8_8192_m1024", (8192, 8192, 8, 8192, 1024, "int8", 8, "mfma"))
bench.WORKLOADS.setdefault("a8w8_4096_m2048", (4096, 4096, 8, 4096, 2048, "int8", 16, "mfma"))
bench.WORKLOADS.setdefault("a8w8_8192_m512", (8192, 8192, 8, 8192, 512, "int8", 8, "mfma"))
bench.WORKLOADS.setdefault("a8w8_4096_m1024", (4096, 4096, 8, 4096, 1024, "int8", 32, "mfma"))
bench.WORKLOADS.setdefault("a8w8_4096_m128", (4096, 4096, 8, 4096, 128, "int8", 32, "mfma"))
bench.WORKLOADS.setdefault("a8w8_4096_m512", (4096, 4096, 8, 4096, 512, "int8", 32, "mfma"))
bench.WORKLOADS.setdefault("a8w8_4096x14336_m256", (4096, 14336, 8, 14336, 256, "int8", 8, "mfma"))
bench.WORKLOADS.setdefault("a8w8_4096x8192_m256", (4096, 8192, 8, 8192, 256, "int8", 16, "mfma"))
bench.WORKLOADS.setdefault("a8w8_2048x8192_m256", (2048, 8192, 8, 8192, 256, "int8", 16, "mfma"))
bench.WORKLOADS.setdefault("a8w8_8192x2048_m128", (8192, 2048, 8, 2048, 128, "int8", 16, "mfma"))
bench.WORKLOADS.setdefault("mx_a8w8_4096x8192_m256", (4096, 8192, 8, 32, 256, "mxa8", 8, "mfma"))
bench.WORKLOADS.setdefault("mx_a8w8_4096_m128", (4096, 4096, 8, 32, 128, "mxa8", 16, "mfma"))
for name in names:
    first = None
    for rep in range(2):
        G = lambda n: (n << 24)
        for t in ((0, 0, 0, 0), (0, 0, 0, 4194304), (0, 0, 0, G(1)), (0, 0, 0, G(2)), (0, 0, 0, G(3))):
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                y = r.call(r.mods[0]).float().cpu().numpy()
                torch.cuda.synchronize()
                if first is None:
                    first = y
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.3)
                print(json.dumps(dict(workload=name, flags=t[3], kernel=r.kernel_name(), chained_us=round(c_us, 3), equal_first=bool(np.array_equal(y, first)),
                                      rel=float(np.abs(y - first).mean() / np.abs(first).mean()))), flush=True)
                del r
            except Exception as e:
                print(json.dumps(dict(workload=name, flags=t[3], error=f"{type(e).__name__}: {e}"[:200])), flush=True)
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
