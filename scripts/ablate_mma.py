"""K-loop ablation of the 8-wave MFMA kernel (needs `make -C gemlite_amd/csrc MMA_EXTRA=-DGL_MMA_EXPERIMENTS`):
EXP bits — 1 barrier + counted wait, 2 dequant VALU (and with it the weight requests), 4 A-fragment reads, 8 x DMA, 16 weight requests."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemlite_amd import GemLiteLinear
from gemlite_amd.core import _hip_matmul
from gemlite_amd.dtypes import TORCH_TO_DTYPE
from gemlite_amd.bench_utils import kernel_device_us
DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)
bf = torch.bfloat16
for tag, N, K, nl, cfgs in (("cfgB", 8192, 8192, 8, ((2, 4), (1, 4), (4, 8))), ("cfgA", 4096, 4096, 16, ((4, 4), (2, 2)))):
    mods = []
    for _ in range(nl):
        W_q = torch.randint(0, 16, (N, K), generator=g, dtype=torch.int32, device=DEV).to(torch.uint8)
        s = (torch.rand(N * K // 128, 1, generator=g, device=DEV) * 0.01 + 0.001).to(bf)
        z = (torch.rand(N * K // 128, 1, generator=g, device=DEV) * 15).to(bf)
        mods.append(GemLiteLinear(4, 128, K, N, TORCH_TO_DTYPE[bf], TORCH_TO_DTYPE[bf]).pack(W_q, s, z, None))
    x = (torch.randn(256, K, generator=g, device=DEV) / 10).to(bf)
    for sk, mi in cfgs:
        row = {}
        for E in (0, 32, 2, 34, 31, 63):
            if mi not in (4, 8) and E:
                continue
            t = (0, sk, mi, E << 8)
            i = [0]

            def launch():
                lin = mods[i[0] % nl]
                i[0] += 1
                return _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, t)
            row[E] = round(kernel_device_us(launch, iters=24, warmup=3), 2)
        print(json.dumps(dict(tag=tag, tile_rows=32 * mi, splitk=sk, us_by_exp=row)), flush=True)
    del mods
    torch.cuda.empty_cache()
