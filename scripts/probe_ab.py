"""Device time of a few headline INT4 shapes with the planner's choice (one line each): the probe of scripts/ab_mma.sh."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemlite_amd import GemLiteLinear
from gemlite_amd.core import _hip_matmul
from gemlite_amd.dtypes import TORCH_TO_DTYPE
from gemlite_amd.bench_utils import kernel_device_us

DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)
CASES = {"cfgAn": (4096, 4096, 4, 256, (0, 0, 32, 0), 64), "cfgA": (4096, 4096, 4, 256, (0, 2, 2, 0), 32), "cfgB": (8192, 8192, 4, 256, (0, 2, 4, 0), 8),
         "pre": (8192, 8192, 4, 2048, (0, 0, 0, 0), 8), "w2": (16384, 16384, 2, 256, (0, 0, 0, 0), 2),
         "m64": (4096, 4096, 4, 64, (0, 0, 0, 0), 32), "m1024": (4096, 4096, 4, 1024, (0, 0, 0, 0), 32)}
for name in (sys.argv[1:] or ["cfgA", "cfgB", "pre"]):
    N, K, nbits, M, tun, nl = CASES[name]
    tdt = torch.bfloat16
    mods = []
    for _ in range(nl):
        W_q = torch.randint(0, 2 ** nbits, (N, K), generator=g, dtype=torch.int32, device=DEV).to(torch.uint8)
        s = (torch.rand(N * K // 128, 1, generator=g, device=DEV) * 0.01 + 0.001).to(tdt)
        z = (torch.rand(N * K // 128, 1, generator=g, device=DEV) * (2 ** nbits - 1)).to(tdt)
        code = TORCH_TO_DTYPE[tdt]
        mods.append(GemLiteLinear(nbits, 128, K, N, code, code).pack(W_q, s, z, None))
        del W_q
    x = (torch.randn(M, K, generator=g, device=DEV) / 10).to(tdt)
    i = [0]

    def launch():
        lin = mods[i[0] % nl]
        i[0] += 1
        return _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tun)
    us = kernel_device_us(launch, iters=60, warmup=6)
    fl = 2.0 * M * N * K
    print(json.dumps(dict(case=name, us=round(us, 2), frac=round(fl / us / 1e6 / 2500, 4))), flush=True)
    del mods
    torch.cuda.empty_cache()
