"""Round 6, late: A8W8 shapes the planner gives to the K-sliced 128-column tiles (gemm_a8w8_lds_kernel) — against the unsplit 64 x 64 / 128 x 128 tiles now that those walk K in
the rotated / grouped order."""
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
SH = {"a8w8_8192_m256": (8192, 8192, 8, 8192, 256, "int8", 8, "mfma"), "fp8_8192_m256": (8192, 8192, 8, 8192, 256, "fp8w8", 8, "mfma"),
      "a8w8_8192_m128": (8192, 8192, 8, 8192, 128, "int8", 8, "mfma"), "a8w8_14336x4096_m256": (14336, 4096, 8, 4096, 256, "int8", 8, "mfma"),
      "a8w8_8192x4096_m256": (8192, 4096, 8, 4096, 256, "int8", 16, "mfma"), "a8w8_4096_m384": (4096, 4096, 8, 4096, 384, "int8", 32, "mfma"),
      "a8w8_11008x4096_m256": (11008, 4096, 8, 4096, 256, "int8", 8, "mfma")}
for m in (512, 768, 1024, 1536, 2048):
    SH[f"a8w8_4096_m{m}"] = (4096, 4096, 8, 4096, m, "int8", 16, "mfma")
    SH[f"fp8_4096_m{m}"] = (4096, 4096, 8, 4096, m, "fp8w8", 16, "mfma")
SH["a8w8_8192_m512"] = (8192, 8192, 8, 8192, 512, "int8", 8, "mfma")
SH["a8w8_8192_m1024"] = (8192, 8192, 8, 8192, 1024, "int8", 8, "mfma")
bench.WORKLOADS.update(SH)
for name in (sys.argv[1:] or list(SH)):
    first = None
    for rep in range(2):
        for t in ((0, 0, 0, 0), (5, 0, 0, 0), (10, 0, 0, 0), (6, 0, 0, 0)):
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                y = r.call(r.mods[0]).float().cpu().numpy()
                torch.cuda.synchronize()
                if first is None:
                    first = y
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.25)
                print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3), rel=float(np.abs(y - first).mean() / np.abs(first).mean()))), flush=True)
                del r
            except Exception as e:
                print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:120])), flush=True)
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
