"""bring-up helper: where the GPU activation quantisers differ from the oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gemlite_amd.quant_utils import scale_activations_nvfp4
from oracle import mx_oracle as MX
M, K = 37, 1024
for tdt in (torch.bfloat16, torch.float32):
    g = torch.Generator().manual_seed(M * 7 + K)
    x = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).to(tdt)
    x[0, :32] = 0
    x[1, 7] = 300.0
    y, s = scale_activations_nvfp4(x.to("cuda:0"))
    yo, so = MX.scale_activations_nvfp4(x.float().numpy())
    yb = y.cpu().numpy()
    d = np.argwhere(yb != yo)
    print(tdt, "mismatching bytes:", len(d))
    xf = x.float().numpy()
    for (m, j) in d[:12]:
        k0 = 2 * j
        blk = k0 // 16
        s8 = so[m, blk]
        full = max(np.float32(MX.fp8_e4m3_decode(np.array([s8], np.uint8))[0]) * np.float32(0.05), np.float32(1e-6))
        q0, q1 = np.float32(xf[m, k0]) / full, np.float32(xf[m, k0 + 1]) / full
        print(f"  m={m} byte={j} gpu={yb[m, j]:#04x} oracle={yo[m, j]:#04x} x=({xf[m,k0]!r},{xf[m,k0+1]!r}) s8={s8} full={full!r} q=({q0!r},{q1!r})")
