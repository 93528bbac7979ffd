"""Round 6, late: the A8W8 int8 planner against forced tile forms over the LLM layer shapes of the planner fixture, M in argv (default 128 256)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
SHAPES = [(1024, 4096), (1536, 8960), (2048, 8192), (2560, 9728), (3072, 8192), (4096, 1024), (4096, 4096), (4096, 11008), (4096, 14336), (5120, 5120), (5120, 13824), (6144, 4096),
          (8192, 2048), (8192, 3072), (8192, 8192), (8960, 1536), (11008, 4096), (12288, 4096), (13824, 5120), (14336, 4096)]
MS = [int(v) for v in sys.argv[1:]] or [128, 256]
DT = os.environ.get("GL_DT", "int8")
for M in MS:
    for (N, K) in SHAPES:
        name = f"a8w8_{DT}_{N}x{K}_m{M}"
        nl = max(2, min(32, int(300e6 // (N * K))))
        bench.WORKLOADS[name] = (N, K, 8, K, M, DT, nl, "mfma")
        rec = dict(M=M, N=N, K=K, us={}, kern={})
        for vn, t in (("auto", (0, 0, 0, 0)), ("sq64", (5, 0, 0, 0)), ("sq128", (10, 0, 0, 0)), ("lds", (6, 0, 0, 0)), ("lds_sk1", (6, 1, 0, 0)), ("lds_sk2", (6, 2, 0, 0)), ("lds_sk3", (6, 3, 0, 0)), ("lds_sk4", (6, 4, 0, 0))):
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.1)
                rec["us"][vn] = round(c_us, 2)
                rec["kern"][vn] = r.kernel_name()
                del r
            except Exception as e:
                rec["us"][vn] = None
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
        print(json.dumps(rec), flush=True)
