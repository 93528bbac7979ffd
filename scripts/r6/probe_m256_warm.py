"""Round 6: how much of the M = 256 launches is HBM-cold weight latency?  The same launch over ONE layer (weights L2 / MALL-warm) against the
bench's rotation over 32 / 8 distinct layers."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip
lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for name in ("a16w4_4096_m256", "a16w4_8192_m256"):
    for nl in (None, 1, 2):
        for label, t in (("wl", None), ("regs", (0, 0, 0, 131072))):
            core.TUNING_OVERRIDE = t
            r = bench.Runner(name, dev, lib, layers=nl)
            c_us, steps, el = r.chained_us_per_launch(min_seconds=0.25)
            print(json.dumps(dict(workload=name, layers=nl or r.layers, path=label, us=round(c_us, 2))), flush=True)
            del r
            core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
