"""Re-time every entry of the shipped tuning table against the planner's own choice (cold weights, one box): entries that no
longer win by > 3 % can be dropped.  Run on the MI355X."""
import os, sys, json, ast
os.environ["GEMLITE_HIP_NO_DEFAULT_CONFIG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemlite_amd import GemLiteLinear
from gemlite_amd.core import _hip_matmul
from gemlite_amd.dtypes import TORCH_TO_DTYPE
from gemlite_amd.bench_utils import kernel_device_us
from tests.test_gpu_parity import _kernel_name
DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)
table = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gemlite_amd", "configs", "mi355x.json")))
cache = {}
for fam, ents in table.items():
    for key, e in ents.items():
        M, N, K, gs, eps, tid = ast.literal_eval(key)
        if (N, K) not in cache:
            cache.clear(); torch.cuda.empty_cache()
            nl = max(2, min(16, int(300e6 // (N * K // 2))))
            mods = []
            for _ in range(nl):
                W_q = torch.randint(0, 16, (N, K), generator=g, dtype=torch.int32, device=DEV).to(torch.uint8)
                s = (torch.rand(N * K // gs, 1, generator=g, device=DEV) * 0.01 + 0.001).half()
                z = (torch.rand(N * K // gs, 1, generator=g, device=DEV) * 15).half()
                mods.append(GemLiteLinear(4, gs, K, N, TORCH_TO_DTYPE[torch.float16], TORCH_TO_DTYPE[torch.float16]).pack(W_q, s, z, None))
                del W_q
            cache[(N, K)] = mods
        mods = cache[(N, K)]
        x = (torch.randn(M, K, generator=g, device=DEV) / 10).half()
        out = {}
        for name, t in (("table", tuple(e["tuning"])), ("planner", (0, 0, 0, 0))):
            i = [0]
            def launch():
                lin = mods[i[0] % len(mods)]; i[0] += 1
                return _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, t)
            us = min(kernel_device_us(launch, iters=20, warmup=3) for _ in range(2))
            out[name] = (round(us, 2), _kernel_name(mods[0], x, -1, t))
        print(json.dumps(dict(fam=fam, key=key, tuning=e["tuning"], **out, gain=round(out["planner"][0] / out["table"][0], 3))), flush=True)
