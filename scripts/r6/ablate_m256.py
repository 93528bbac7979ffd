"""Round 6: K-loop ablation of the two M = 256 headline kernels (needs `make -C gemlite_amd/csrc MMA_EXTRA=-DGL_MMA_EXPERIMENTS`).
EXP bits (tuning[3] >> 20): 1 barrier + counted wait | 2 dequant VALU | 4 A-fragment reads | 8 x DMA | 16 weight requests.  Results of EXP != 0
are wrong by construction; what is measured is the time each part of the loop is responsible for — the CEILING of any rewrite of it."""
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
    for E in (0, 2, 4, 6, 8, 1, 18, 30, 31):
        core.TUNING_OVERRIDE = (0, 0, 0, E << 20)
        try:
            r = bench.Runner(name, dev, lib)
            c_us, steps, el = r.chained_us_per_launch(min_seconds=0.25)
            print(json.dumps(dict(workload=name, exp=E, kernel=r.kernel_name(), us=round(c_us, 2))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, exp=E, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
