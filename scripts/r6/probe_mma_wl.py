"""Round 6: the M = 256 tile kernels with the packed words through LDS (WL, default) against the round-5 register path (tuning[3] & 131072), with an
on-the-spot comparison of the two outputs (they must be bit-identical: same arithmetic, same order)."""
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
bench.WORKLOADS.update({"a16w4_4096_m128": (4096, 4096, 4, 128, 128, "bf16", 32, "mfma"), "a16w4_4096_m192": (4096, 4096, 4, 128, 192, "bf16", 32, "mfma"),
                        "a16w4_8192_m128": (8192, 8192, 4, 128, 128, "bf16", 8, "mfma"), "a16w4_4096_m256_f16": (4096, 4096, 4, 128, 256, "fp16", 32, "mfma"),
                        "a16w4_11008_m256": (11008, 4096, 4, 128, 256, "bf16", 12, "mfma"), "a16w4_4096x11008_m256": (4096, 11008, 4, 128, 256, "bf16", 12, "mfma")})
for name in sys.argv[1:] or ["a16w4_4096_m256", "a16w4_8192_m256", "a16w4_4096_m128", "a16w4_4096_m192", "a16w4_8192_m128", "a16w4_4096_m256_f16", "a16w4_11008_m256", "a16w4_4096x11008_m256", "a16w4_8192_m2048"]:
    ref = None
    for label, t in (("wl", (0, 0, 0, 0)), ("regs", (0, 0, 0, 131072)), ("wl_r5_epilogue", (0, 0, 0, 262144)), ("wl", (0, 0, 0, 0)), ("r5", (0, 0, 0, 131072 + 262144))):
        core.TUNING_OVERRIDE = t if any(t) else None
        try:
            r = bench.Runner(name, dev, lib, layers=4 if "2048" in name else None)
            y = r.call(r.mods[0]).float().cpu().numpy()
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            c_us, steps, el = r.chained_us_per_launch(min_seconds=0.25)
            print(json.dumps(dict(workload=name, path=label, kernel=r.kernel_name(), us=round(c_us, 2), bitwise_equal_first=bool(np.array_equal(y, ref)),
                                  finite=bool(np.isfinite(y).all()))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, path=label, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
