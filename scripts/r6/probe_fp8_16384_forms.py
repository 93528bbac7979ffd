import json, os, sys
import torch
sys.path.insert(0, os.getcwd())
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip
lib = _hip.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
G = lambda n: (n << 24)
for name in ("fp8_16384_m256",):
    for rep in range(2):
        for t in ((0,0,0,0), (5,0,0,0), (5,0,0,G(1)), (5,0,0,G(2)), (5,0,4,G(1)), (5,0,0,4194304), (6,0,0,0)):
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.3)
                print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3))), flush=True)
                del r
            except Exception as e:
                print(json.dumps(dict(workload=name, tuning=t, error=str(e)[:100])), flush=True)
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
