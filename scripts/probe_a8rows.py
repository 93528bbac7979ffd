import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from gemlite_amd.core import _hip_matmul
from gemlite_amd.bench_utils import kernel_device_us
from gemlite_amd.helper import A8W8_int8_dynamic, A8W8_fp8_dynamic
from gemlite_amd.quant_utils import scale_activations_per_token
from tests.test_gpu_parity import _kernel_name
DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(1)
for tag, cls, qdt, N, K, nl in (("int8 8192", A8W8_int8_dynamic, torch.int8, 8192, 8192, 4), ("int8 14336x4096", A8W8_int8_dynamic, torch.int8, 14336, 4096, 4),
                                ("int8 4096x14336", A8W8_int8_dynamic, torch.int8, 4096, 14336, 4), ("fp8 16384", A8W8_fp8_dynamic, torch.float8_e4m3fn, 16384, 16384, 2)):
    proc = cls(device=DEV, dtype=torch.float16)
    mods = [proc.from_weights((torch.randn(N, K, generator=g, device=DEV) / 30).half()) for _ in range(nl)]
    for M in (2, 8, 16):
        x = (torch.randn(M, K, generator=g, device=DEV) / 10).half()
        xq, sx = scale_activations_per_token(x, qdt)
        res = {}
        for t in ((0, 0, 0, 0), (0, 0, 1, 0), (1, 0, 0, 0)):
            i = [0]
            def launch():
                lin = mods[i[0] % nl]; i[0] += 1
                return _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, t)
            us = kernel_device_us(launch, iters=20, warmup=3)
            res[_kernel_name(mods[0], xq, -1, t)] = round(us, 2)
        print(json.dumps(dict(tag=tag, M=M, res=res)), flush=True)
    del mods
    torch.cuda.empty_cache()
