"""A8W8 8-wave kernels: weights through LDS (default for 128- / 256-row tiles) vs straight from memory (tuning[3] & 64)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemlite_amd.core import _hip_matmul
from gemlite_amd.bench_utils import kernel_device_us
from gemlite_amd.helper import A8W8_int8_dynamic, A8W8_fp8_dynamic
from gemlite_amd.quant_utils import scale_activations_per_token
from tests.test_gpu_parity import _kernel_name
DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(1)
for tag, cls, qdt, N, K, nl, Ms in (("int8 4096", A8W8_int8_dynamic, torch.int8, 4096, 4096, 16, (256, 512, 1024, 4096)),
                                   ("fp8 8192", A8W8_fp8_dynamic, torch.float8_e4m3fn, 8192, 8192, 4, (256, 1024, 2048)),
                                   ("fp8 16384", A8W8_fp8_dynamic, torch.float8_e4m3fn, 16384, 16384, 2, (256, 1024))):
    proc = cls(device=DEV, dtype=torch.float16)
    mods = [proc.from_weights((torch.randn(N, K, generator=g, device=DEV) / 30).half()) for _ in range(nl)]
    for M in Ms:
        x = (torch.randn(M, K, generator=g, device=DEV) / 10).half()
        xq, sx = scale_activations_per_token(x, qdt)
        for t in ((0, 0, 0, 0), (0, 0, 0, 64), (0, 0, 4, 0), (0, 0, 8, 0), (0, 1, 8, 0), (0, 2, 8, 0), (0, 4, 8, 0), (0, 2, 4, 0)):
            i = [0]
            def launch():
                lin = mods[i[0] % nl]; i[0] += 1
                return _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, t)
            try:
                us = kernel_device_us(launch, iters=20, warmup=3)
                name = _kernel_name(mods[0], xq, -1, t)
            except Exception as e:
                print(json.dumps(dict(tag=tag, M=M, tuning=t, error=str(e)[:60])), flush=True); continue
            ops = 2.0 * M * N * K
            print(json.dumps(dict(tag=tag, M=M, tuning=t, kernel=name, us=round(us, 2), frac=round(ops / us / 1e6 / 5000, 3))), flush=True)
    del mods
    torch.cuda.empty_cache()
