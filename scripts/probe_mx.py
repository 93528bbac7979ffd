"""device time of the block-scaled MFMA kernels across tile heights / K slices (needs the GPU): which (tuning[1], tuning[2]) wins
per shape, next to the planner's default.  Output: one JSON line per (processor, N, K, M)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemlite_amd import core as C, helper as H
from gemlite_amd.bench_utils import kernel_device_us
from gemlite_amd.quant_utils import scale_activations_mxfp4, scale_activations_mxfp8

dev = "cuda:0"
tdt = torch.bfloat16
PROCS = {"a8w8": lambda: H.A8W8_MXFP_dynamic(device=dev, dtype=tdt, post_scale=False),
         "a8w4": lambda: H.A8W4_MXFP_dynamic(device=dev, dtype=tdt, post_scale=False),
         "a4w4": lambda: H.A4W4_MXFP_dynamic(device=dev, dtype=tdt),
         "a16w4": lambda: H.A16W4_MXFP(device=dev, dtype=tdt)}
shapes = [(4096, 4096), (8192, 8192), (14336, 4096), (4096, 14336)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for pname, mk in PROCS.items():
    for (N, K) in shapes:
        lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=tdt)
        layer = mk().from_linear(lin, del_orig=True)
        meta = layer.get_meta_args()
        for M in (8, 32, 64, 128, 256, 512):
            x = (torch.randn(M, K, device=dev) / 10).to(tdt)
            sx = None
            if pname in ("a8w8", "a8w4"):
                x, sx = scale_activations_mxfp8(x)
            elif pname == "a4w4":
                x, sx = scale_activations_mxfp4(x)
            res = {}
            cands = [(0, 0, 0, 0)] + [(2, sk, mi, 0) for mi in (1, 2, 4) + ((8,) if pname == "a16w4" else ()) for sk in (1, 2, 3, 4, 6, 8)]
            if pname in ("a8w8", "a4w4"):
                cands.append((3, 0, 0, 0))
            for cand in cands:
                try:
                    us = kernel_device_us(lambda: C._hip_matmul(x, layer.W_q, layer.scales, layer.zeros, sx, meta, -1, cand),
                                          iters=12, before_each=lambda: flush.fill_(1))
                except Exception:
                    continue
                if us == us:
                    res[str(cand)] = round(us, 2)
            best = min(res, key=res.get)
            print(json.dumps({"proc": pname, "N": N, "K": K, "M": M, "default": res.get("(0, 0, 0, 0)"), "best": best, "best_us": res[best],
                              "all": res}), flush=True)
        del layer
        torch.cuda.empty_cache()
