"""GEMV (M = 1) variants: per-launch event time and chained (hipGraph) time per launch for the decode shapes.
    gpurun -- 'bash scripts/gpu.sh probe:probe_gemv.py'"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
CASES = {
    "a16w4_4096_m1": [(0, 0, 0, 0), (2, 1, 8, 0), (2, 1, 4, 0), (3, 2, 0, 0), (3, 2, 8, 0), (4, 4, 0, 0)],
    "a16w4_8192_m1": [(0, 0, 0, 0), (0, 0, 8, 0), (4, 2, 8, 0), (4, 2, 0, 0), (3, 2, 8, 0)],
    "a16w4_16384_m1": [(0, 0, 0, 0), (0, 0, 8, 0), (3, 1, 8, 0), (3, 1, 0, 0)],
    "a16w2_16384_m1": [(0, 0, 0, 0), (0, 0, 8, 0)],
    "a16w4_11008_m1": [(0, 0, 0, 0), (3, 1, 0, 0), (3, 1, 8, 0)],
    "a8w8_4096_m1": [(0, 0, 0, 0)],
}
for name, tunings in CASES.items():
    if sys.argv[1:] and name not in sys.argv[1:]:
        continue
    for t in tunings:
        core.TUNING_OVERRIDE = t
        try:
            r = bench.Runner(name, dev, lib)
            out = r.roofline(96)
            print(json.dumps(dict(workload=name, tuning=t, kernel=out["kernel"], kernel_us=out["kernel_us"],
                                  chained_us=out["us_per_launch_chained"], frac=out["frac"], gbs=out["achieved"],
                                  gbs_chained=out["gap_inclusive"])), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:120])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
