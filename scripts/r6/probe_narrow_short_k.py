import json, os, sys
import torch
sys.path.insert(0, os.getcwd())
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip
lib = _hip.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
bench.WORKLOADS["a16w4_4096x1024_m128"] = (4096, 1024, 4, 128, 128, "bf16", 32, "mfma")
bench.WORKLOADS["a16w4_4096x1024_m100"] = (4096, 1024, 4, 128, 100, "bf16", 32, "mfma")
bench.WORKLOADS["a16w4_4096x1536_m128"] = (4096, 1536, 4, 128, 128, "bf16", 32, "mfma")
bench.WORKLOADS["a16w2_4096x1024_m128"] = (4096, 1024, 2, 128, 128, "bf16", 32, "mfma")
for nb in (4, 2):
    for m in (96, 128, 192, 256):
        bench.WORKLOADS[f"a16w{nb}_1024x4096_m{m}"] = (1024, 4096, nb, 128, m, "bf16", 32, "mfma")
bench.WORKLOADS["a16w4_1024x8192_m128"] = (1024, 8192, 4, 128, 128, "bf16", 32, "mfma")
bench.WORKLOADS["a16w4_512x4096_m256"] = (512, 4096, 4, 128, 256, "bf16", 32, "mfma")
NAMES = sys.argv[1:] or ["a16w4_4096x1024_m128", "a16w4_4096x1024_m100", "a16w4_4096x1536_m128", "a16w2_4096x1024_m128"]
for name in NAMES:
    for rep in range(2):
        for t in ((0,0,0,0), (0,0,0,16384)):
            core.TUNING_OVERRIDE = t if any(t) else None
            r = bench.Runner(name, dev, lib)
            c_us, n, el = r.chained_us_per_launch(min_seconds=0.2)
            print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3))), flush=True)
            del r
            core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
