"""Round 6, late: the block-scaled (MX) tile planner against forced tile forms over the LLM layer shapes of the planner fixture.
GL_DT = mxa8 (default) | mxa4, GL_WB = 8 (default) | 4; M values in argv (default 128 256)."""
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
if os.environ.get("GL_LONG_K"):  # only the layers with K > 8192
    SHAPES = [sh for sh in SHAPES if sh[1] > 8192 and sh[1] % (512 if os.environ.get("GL_DT") == "mxa4" else 256) == 0]
MS = [int(v) for v in sys.argv[1:]] or [128, 256]
DT = os.environ.get("GL_DT", "mxa8")
WB = int(os.environ.get("GL_WB", "8"))
VARIANTS = (("auto", (0, 0, 0, 0)), ("sq", (6, 0, 0, 0)), ("m32", (2, 0, 1, 0)), ("m64", (2, 0, 2, 0)), ("m128", (2, 0, 4, 0)), ("m64_sk1", (2, 1, 2, 0)), ("m64_sk2", (2, 2, 2, 0)),
            ("m128_sk1", (2, 1, 4, 0)), ("m128_sk2", (2, 2, 4, 0)), ("m128_sk3", (2, 3, 4, 0)), ("m128_sk4", (2, 4, 4, 0)))
if os.environ.get("GL_FEW"):  # auto / unsplit 64 x 64 / 128-row tiles only
    VARIANTS = VARIANTS[:2] + VARIANTS[4:5]
for M in MS:
    for (N, K) in SHAPES:
        name = f"mx_{DT}_w{WB}_{N}x{K}_m{M}"
        nl = max(2, min(32, int(300e6 // (N * K))))
        bench.WORKLOADS[name] = (N, K, WB, 32, M, DT, nl, "mfma")
        rec = dict(M=M, N=N, K=K, us={}, kern={})
        for vn, t in VARIANTS:
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
