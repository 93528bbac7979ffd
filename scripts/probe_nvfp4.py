"""Round 4: the NVFP4 layer (A4W4_NVFP_dynamic) on the fp16 MFMA tile kernel (x expanded in front, weights in the K loop) against the
coverage kernel it ran on before (tuning[0] = 1): graph-replayed time of the matmul call on pre-quantised activations, and of layer(x).
    python scripts/probe_nvfp4.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
import gemlite_amd.core as core
from gemlite_amd import _hip
from gemlite_amd.core import _hip_matmul
from gemlite_amd.quant_utils import scale_activations_nvfp4, scale_activations_mxfp4

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
H = gemlite_amd.helper


def graph_us(fn, n_inner, min_seconds=0.15):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n_inner):
                fn(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < min_seconds:
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize(); reps += 5
        el = time.perf_counter() - t0
    torch.cuda.current_stream().wait_stream(s)
    return el / (reps * n_inner) * 1e6


for N, K, nl in ((4096, 4096, 16), (8192, 8192, 4)):
    for procname in ("A4W4_NVFP_dynamic", "A4W4_MXFP_dynamic"):
        layers = []
        for i in range(nl):
            lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=torch.bfloat16)
            lin.weight.data /= 10.0
            layers.append(getattr(H, procname)(device=dev, dtype=torch.bfloat16).from_linear(lin, del_orig=True))
        quant = scale_activations_nvfp4 if "NVFP" in procname else scale_activations_mxfp4
        for M in (1, 16, 64, 256, 1024):
            x = (torch.randn(M, K, device=dev) / 4).to(torch.bfloat16)
            xq, sx = quant(x)
            for tuning in ((0, 0, 0, 0), (1, 0, 0, 0)):
                if tuning[0] == 1 and (M > 64 or "MXFP" in procname):
                    continue
                def mm(i):
                    l = layers[i % nl]
                    return _hip_matmul(xq, l.W_q, l.scales, l.zeros, sx, l.get_meta_args(), -1, tuning)
                y = mm(0)
                torch.cuda.synchronize()
                a = core._static_args(layers[0].W_q, layers[0].scales, layers[0].zeros, layers[0].get_meta_args())
                a.matmul_type, a.M, a.x, a.out, a.scales_x = -1, M, xq.data_ptr(), y.data_ptr(), sx.data_ptr()
                a.input_dtype = layers[0].input_dtype.value
                a.stride_xm, a.stride_xk, a.stride_om, a.stride_on, a.stride_sx_m = xq.stride(0), 1, N, 1, sx.stride(0)
                for i in range(4):
                    a.tuning[i] = tuning[i]
                name = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
                us = graph_us(mm, max(nl, 8))
                rec = dict(proc=procname, N=N, K=K, M=M, tuning=tuning, kernel=name, matmul_us=round(us, 2))
                if tuning[0] == 0:
                    rec["layer_e2e_us"] = round(graph_us(lambda i: layers[i % nl](x), max(nl, 8)), 2)
                print(json.dumps(rec), flush=True)
        del layers
        torch.cuda.empty_cache()
