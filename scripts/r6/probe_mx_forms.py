"""Round 6, late: block-scaled shapes around the hand-over from the unsplit 64 x 64 tiles (gemm_mx_sq_kernel, now with the rotated / grouped K order) to the 128-column tiles."""
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
SH = {"mx_a8w8_8192_m256": (8192, 8192, 8, 32, 256, "mxa8", 4, "mfma"), "mx_a4w4_8192_m256": (8192, 8192, 4, 32, 256, "mxa4", 8, "mfma"),
      "mx_a8w8_4096_m512": (4096, 4096, 8, 32, 512, "mxa8", 16, "mfma"), "mx_a8w8_4096_m1024": (4096, 4096, 8, 32, 1024, "mxa8", 16, "mfma"),
      "mx_a8w8_14336x4096_m256": (14336, 4096, 8, 32, 256, "mxa8", 4, "mfma"), "mx_a8w8_4096x14336_m256": (4096, 14336, 8, 32, 256, "mxa8", 4, "mfma"),
      "mx_a4w4_4096_m512": (4096, 4096, 4, 32, 512, "mxa4", 16, "mfma")}
SH.update({"mx_a8w8_4096_m768": (4096, 4096, 8, 32, 768, "mxa8", 16, "mfma"), "mx_a8w8_4096_m2048": (4096, 4096, 8, 32, 2048, "mxa8", 8, "mfma"),
           "mx_a8w8_8192_m512": (8192, 8192, 8, 32, 512, "mxa8", 4, "mfma"), "mx_a8w8_8192_m1024": (8192, 8192, 8, 32, 1024, "mxa8", 4, "mfma"),
           "mx_a4w4_4096_m1024": (4096, 4096, 4, 32, 1024, "mxa4", 16, "mfma"), "mx_a4w4_4096_m2048": (4096, 4096, 4, 32, 2048, "mxa4", 8, "mfma"),
           "mx_a4w4_8192_m512": (8192, 8192, 4, 32, 512, "mxa4", 8, "mfma"), "mx_a4w4_8192_m1024": (8192, 8192, 4, 32, 1024, "mxa4", 8, "mfma"),
           "mx_a8w8_4096x14336_m512": (4096, 14336, 8, 32, 512, "mxa8", 4, "mfma")})
SH.update({"mx_a8w4_4096_m512": (4096, 4096, 4, 32, 512, "mxa8", 16, "mfma"), "mx_a8w4_4096_m1024": (4096, 4096, 4, 32, 1024, "mxa8", 16, "mfma"),
           "mx_a8w4_8192_m512": (8192, 8192, 4, 32, 512, "mxa8", 8, "mfma"), "mx_a8w4_8192_m1024": (8192, 8192, 4, 32, 1024, "mxa8", 8, "mfma"), "mx_a8w4_4096_m2048": (4096, 4096, 4, 32, 2048, "mxa8", 8, "mfma")})
bench.WORKLOADS.update(SH)
for name in (sys.argv[1:] or list(SH)):
    for rep in range(2):
        for t in ((0, 0, 0, 0), (6, 0, 0, 0), (2, 0, 0, 0), (3, 0, 0, 0)):
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.25)
                print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3))), flush=True)
                del r
            except Exception as e:
                print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:120])), flush=True)
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
