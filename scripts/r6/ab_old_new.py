"""A/B of two library builds on one box: python scripts/r6/ab_old_new.py <repo dir> <workload> ...   (prints chained us per launch)"""
import json, os, sys
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
os.chdir(root)
import torch
import bench
from gemlite_amd import _hip
lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for name in sys.argv[2:]:
    r = bench.Runner(name, dev, lib)
    c_us, n, el = r.chained_us_per_launch(min_seconds=0.4)
    print(json.dumps(dict(build=os.path.basename(root) or "new", workload=name, kernel=r.kernel_name(), chained_us=round(c_us, 3))), flush=True)
    del r
    torch.cuda.empty_cache()
