"""Round 6, late: the tile planner of the packed 2-bit (GL_BITS=2, default) / 4-bit words under fp16 activations against forced tile forms over the LLM layer
shapes of the planner fixture (groups of 128).  M values in argv (default 128 256)."""
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
if os.environ.get("GL_SHAPE"):  # e.g. 16384x16384
    SHAPES = [tuple(int(v) for v in sh.split("x")) for sh in os.environ["GL_SHAPE"].split(",")]
MS = [int(v) for v in sys.argv[1:]] or [128, 256]
BITS = int(os.environ.get("GL_BITS", "2"))
VARIANTS = (("auto", (0, 0, 0, 0)), ("n1", (0, 1, 32, 0)), ("n2", (0, 2, 32, 0)), ("n4", (0, 4, 32, 0)), ("w64", (0, 0, 2, 0)), ("w128", (0, 0, 4, 0)), ("w256", (0, 0, 8, 0)),
            ("w128_sk1", (0, 1, 4, 0)), ("w128_sk2", (0, 2, 4, 0)), ("w128_sk3", (0, 3, 4, 0)), ("w256_sk1", (0, 1, 8, 0)), ("w256_sk2", (0, 2, 8, 0)))
if os.environ.get("GL_VARIANTS"):  # e.g. auto,n1,w64,w128,w128_sk1,w256_sk1,w256_sk2
    VARIANTS = tuple(v for v in VARIANTS if v[0] in os.environ["GL_VARIANTS"].split(","))
for M in MS:
    for (N, K) in SHAPES:
        name = f"a16w{BITS}_{N}x{K}_m{M}"
        nl = max(2, min(12, int(2.4e9 // (N * K * BITS))))
        bench.WORKLOADS[name] = (N, K, BITS, int(os.environ.get("GL_GROUP", "128")), M, os.environ.get("GL_DT", "fp16"), nl, "mfma")
        rec = dict(M=M, N=N, K=K, us={}, kern={})
        for vn, t in VARIANTS:
            if vn in ("w256", "w256_sk1", "w256_sk2") and M <= 128:
                continue
            core.TUNING_OVERRIDE = t if any(t) else None
            try:
                r = bench.Runner(name, dev, lib)
                c_us, n, el = r.chained_us_per_launch(min_seconds=0.08)
                if c_us < 1000:
                    rec["us"][vn] = round(c_us, 2)
                    rec["kern"][vn] = r.kernel_name()
                del r
            except Exception as e:
                rec["us"][vn] = None
            finally:
                core.TUNING_OVERRIDE = None
            torch.cuda.empty_cache()
        print(json.dumps(rec), flush=True)
