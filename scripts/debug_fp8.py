"""Step-by-step run of the FP8 x FP8 16384^2 paths with a sync + print after every launch (stderr visible)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
from gemlite_amd.core import _hip_matmul
from gemlite_amd.quant_utils import scale_activations_per_token
DEV = "cuda:0"
N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator().manual_seed(7)
W = (torch.randn(N, K, generator=g) / 30).half()
lin = gemlite_amd.helper.A8W8_fp8_dynamic(device=DEV, dtype=torch.float16).from_weights(W)
del W
print("layer built", flush=True)
for M in (1, 256):
    x = (torch.randn(M, K, generator=g) / 10).half().to(DEV)
    y = lin(x); torch.cuda.synchronize(); print(f"M={M} lin(x) ok", float(y.float().abs().mean()), flush=True)
    xq, sx = scale_activations_per_token(x, torch.float8_e4m3fn); torch.cuda.synchronize(); print("  act quant ok", flush=True)
    for t in ((1, 0, 0, 0), (2, 0, 0, 0), (0, 0, 0, 0)):
        y2 = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, t)
        torch.cuda.synchronize()
        print(f"  tuning {t} ok, max diff vs lin(x) {float((y.float() - y2.float()).abs().max()):.4g}", flush=True)
print("done")
