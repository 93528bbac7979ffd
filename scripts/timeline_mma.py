"""s_memtime timeline of block (0,0) of the 8-wave MFMA kernel (tuning[3] & 4): where a block spends its cycles.
Stamps per wave: 0 start | 1 prologue barrier passed | 2 first step done | 3 main loop done, requests retired |
4 K halves added | 5 tile staged / slab stores issued | 6 ticket taken | 7 last arriver done."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemlite_amd import GemLiteLinear, _hip
from gemlite_amd.core import _hip_matmul
from gemlite_amd.dtypes import TORCH_TO_DTYPE
from gemlite_amd.bench_utils import kernel_device_us

DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)
PROBE0 = (65536 - 4096) * 4


def run(tag, N, K, M, tdt, tuning, nl=8):
    mods = []
    for _ in range(nl):
        W_q = torch.randint(0, 16, (N, K), generator=g, dtype=torch.int32, device=DEV).to(torch.uint8)
        s = (torch.rand(N * K // 128, 1, generator=g, device=DEV) * 0.01 + 0.001).to(tdt)
        z = (torch.rand(N * K // 128, 1, generator=g, device=DEV) * 15).to(tdt)
        code = TORCH_TO_DTYPE[tdt]
        mods.append(GemLiteLinear(4, 128, K, N, code, code).pack(W_q, s, z, None))
    x = (torch.randn(M, K, generator=g, device=DEV) / 10).to(tdt)
    t = tuple(tuning[:3]) + (tuning[3] | 4,)
    rows = []
    for i in range(3 * nl):
        lin = mods[i % nl]
        _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, t)
        torch.cuda.synchronize()
        ws = list(_hip._workspaces.values())[0]
        st = ws[PROBE0: PROBE0 + 8 * 16 * 8].view(torch.int64).cpu().numpy().reshape(8, 16)
        if i >= nl:
            rows.append(st[:, :8] - st[:, :1].min())
    import numpy as np
    m = np.mean(np.stack(rows), axis=0)
    i = [0]

    def launch():
        lin = mods[i[0] % nl]
        i[0] += 1
        return _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tuning)
    us = kernel_device_us(launch, iters=30, warmup=3)
    print(json.dumps(dict(tag=tag, tuning=list(tuning), kernel_us=round(us, 2),
                          mean_over_waves=[int(v) for v in m.mean(axis=0)], slowest_wave=[int(v) for v in m.max(axis=0)],
                          wave0=[int(v) for v in m[0]], wave4=[int(v) for v in m[4]])), flush=True)


bf = torch.bfloat16
for tag, N, K, M, tun in (("cfgA 256x128 auto", 4096, 4096, 256, (0, 0, 0, 0)), ("cfgA 128x128 sk4", 4096, 4096, 256, (0, 4, 4, 0)),
                          ("cfgA 128x128 sk4 noxcd", 4096, 4096, 256, (0, 4, 4, 8)), ("cfgA 256x128 sk4", 4096, 4096, 256, (0, 4, 8, 0)),
                          ("cfgB 256x128 sk4", 8192, 8192, 256, (0, 4, 8, 0)), ("cfgB 256x128 sk4 noxcd", 8192, 8192, 256, (0, 4, 8, 8)),
                          ("cfgB 128x128 sk2", 8192, 8192, 256, (0, 2, 4, 0)), ("cfgB 256x128 sk2", 8192, 8192, 256, (0, 2, 8, 0))):
    run(tag, N, K, M, bf, tun, nl=8 if K > 4096 else 16)
