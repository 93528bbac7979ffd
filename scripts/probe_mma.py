"""Device time of the 8-wave MFMA kernel (gemm_wn_mma) across tile heights / split-K factors, next to the round-1
tiled kernel (tuning[0] = 2) and the few-row kernels; one line per configuration.  Run on the MI355X:
    gpurun -- 'bash scripts/gpu.sh probe:probe_mma.py'"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemlite_amd import GemLiteLinear, _hip
from gemlite_amd.core import _hip_matmul
from gemlite_amd.dtypes import TORCH_TO_DTYPE
from gemlite_amd.bench_utils import kernel_device_us
from tests.test_gpu_parity import _kernel_name  # noqa

DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)


def layers(N, K, nbits, gs, tdt, n):
    out = []
    for _ in range(n):
        W_q = torch.randint(0, 2 ** nbits, (N, K), generator=g, dtype=torch.int32, device=DEV).to(torch.uint8)
        s = (torch.rand(N * K // gs, 1, generator=g, device=DEV) * 0.01 + 0.001).to(tdt)
        z = (torch.rand(N * K // gs, 1, generator=g, device=DEV) * (2 ** nbits - 1)).to(tdt)
        code = TORCH_TO_DTYPE[tdt]
        out.append(GemLiteLinear(nbits, gs, K, N, code, code).pack(W_q, s, z, None))
    return out


def run(tag, N, K, nbits, M, tdt, tunings, nl=8, gs=128, mt=-1):
    mods = layers(N, K, nbits, gs, tdt, nl)
    x = (torch.randn(M, K, generator=g, device=DEV) / 10).to(tdt)
    flops = 2.0 * M * N * K
    for t in tunings:
        i = [0]

        def launch():
            lin = mods[i[0] % nl]
            i[0] += 1
            return _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), mt, t)
        try:
            us = kernel_device_us(launch, iters=40, warmup=4)
            name = _kernel_name(mods[0], x, mt, t)
        except Exception as e:
            print(json.dumps(dict(tag=tag, tuning=t, error=str(e)[:80])), flush=True)
            continue
        print(json.dumps(dict(tag=tag, M=M, tuning=t, kernel=name, us=round(us, 2), tflops=round(flops / us / 1e6, 1),
                              frac=round(flops / us / 1e6 / 2500, 3))), flush=True)
    del mods
    torch.cuda.empty_cache()


bf, hf = torch.bfloat16, torch.float16
which = sys.argv[1:] or ["cfgA", "cfgB", "a8", "rows"]
if "cfgA" in which:
    run("cfgA bf16", 4096, 4096, 4, 256, bf, [(0, 0, 0, 0), (0, 4, 4, 0), (0, 2, 2, 0), (0, 1, 1, 0), (0, 4, 8, 0), (2, 0, 0, 0)], nl=32)
if "cfgB" in which:
    run("cfgB bf16", 8192, 8192, 4, 256, bf, [(0, 0, 0, 0), (0, 2, 4, 0), (0, 1, 2, 0), (0, 3, 8, 0), (0, 4, 8, 0), (2, 0, 0, 0)], nl=8)
if "rows" in which:
    for M in (64, 128, 512):
        run(f"4096 bf16 M={M}", 4096, 4096, 4, M, bf, [(0, 0, 0, 0), (0, 0, 1, 0), (0, 0, 2, 0), (0, 0, 4, 0), (0, 0, 8, 0), (2, 0, 0, 0)], nl=16)
if "w2" in which:
    run("A16W2 16384 bf16", 16384, 16384, 2, 256, bf, [(0, 0, 0, 0), (0, 1, 4, 0)], nl=2)
if "oddk" in which:
    for M in (1, 8, 32, 64):
        run(f"K=11008 fp16 M={M}", 4096, 11008, 4, M, hf, [(0, 0, 0, 0)], nl=12)
if "a8" in which or "a8s" in which:
    from gemlite_amd.helper import A8W8_int8_dynamic, A8W8_fp8_dynamic
    from gemlite_amd.quant_utils import scale_activations_per_token
    for tag, proc_cls, qdt, N, K, nl in (("A8W8 int8 4096", A8W8_int8_dynamic, torch.int8, 4096, 4096, 16),
                                         ("FP8 16384", A8W8_fp8_dynamic, torch.float8_e4m3fn, 16384, 16384, 2))[:1 if "a8s" in which else 2]:
        gg = torch.Generator().manual_seed(1)
        proc = proc_cls(device=DEV, dtype=torch.float16)
        mods = [proc.from_weights((torch.randn(N, K, generator=gg) / 30).half()) for _ in range(nl)]
        for M in ((1, 16, 64, 256, 1024) if N == 4096 else (1, 256)):
            x = (torch.randn(M, K, generator=g, device=DEV) / 10).half()
            xq, sx = scale_activations_per_token(x, qdt)
            ops = 2.0 * M * N * K
            tun = [(0, 0, 0, 0), (0, 0, 0, 32), (2, 0, 0, 0)] + ([(1, 0, 0, 0)] if M <= 16 else []) + ([(0, 2, 2, 0), (0, 2, 4, 0), (0, 4, 8, 0), (0, 4, 4, 0), (0, 1, 1, 0)] if M == 256 else [])
            for t in tun:
                i = [0]

                def launch():
                    lin = mods[i[0] % nl]
                    i[0] += 1
                    return _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, t)
                try:
                    us = kernel_device_us(launch, iters=30, warmup=3)
                    from tests.test_gpu_parity import _kernel_name as kn
                    name = kn(mods[0], xq, -1, t)
                except Exception as e:
                    print(json.dumps(dict(tag=tag, M=M, tuning=t, error=str(e)[:80])), flush=True)
                    continue
                print(json.dumps(dict(tag=tag, M=M, tuning=t, kernel=name, us=round(us, 2), tops=round(ops / us / 1e6, 1),
                                      frac_int8=round(ops / us / 1e6 / 5000, 3), gbs=round((N * K + M * K) / us / 1e3, 1))), flush=True)
        del mods
        torch.cuda.empty_cache()
