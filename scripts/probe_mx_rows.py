"""Round 4: the few-row scaled-MFMA kernel (mx_rows_kernel, tuning[0] = 4) against the 8-wave tile kernel (tuning[0] = 2) and the streaming
kernel (default up to 4 rows) for the MX layers: graph-replayed time of the matmul on pre-quantised activations, and of layer(x).
    python scripts/probe_mx_rows.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
import gemlite_amd.core as core
from gemlite_amd import _hip
from gemlite_amd.core import _hip_matmul
from gemlite_amd.quant_utils import scale_activations_mxfp4, scale_activations_mxfp8

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
H = gemlite_amd.helper


def graph_us(fn, n_inner, min_seconds=0.12):
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
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < min_seconds:
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize(); reps += 5
        el = time.perf_counter() - t0
    torch.cuda.current_stream().wait_stream(s)
    return el / (reps * n_inner) * 1e6


for N, K, nl in ((4096, 4096, 16), (4096, 11008, 8)):
    for procname, quant in (("A8W8_MXFP_dynamic", scale_activations_mxfp8), ("A8W4_MXFP_dynamic", scale_activations_mxfp8), ("A4W4_MXFP_dynamic", scale_activations_mxfp4)):
        layers = []
        for i in range(nl):
            lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=torch.bfloat16)
            lin.weight.data /= 10.0
            kw = {} if procname.startswith("A4") else dict(post_scale=False)  # block scales for x (post_scale=True: one fp32 scale per token)
            layers.append(getattr(H, procname)(device=dev, dtype=torch.bfloat16, **kw).from_linear(lin, del_orig=True))
        for M in (1, 4, 16, 32, 64):
            x = (torch.randn(M, K, device=dev) / 4).to(torch.bfloat16)
            xq, sx = quant(x)
            rec = dict(proc=procname, N=N, K=K, M=M)
            for label, tuning in (("rows", (4, 0, 0, 0)), ("tile", (2, 0, 0, 0)), ("gemv", (5, 0, 0, 0)), ("default", (0, 0, 0, 0))):
                def mm(i):
                    l = layers[i % nl]
                    return _hip_matmul(xq, l.W_q, l.scales, l.zeros, sx, l.get_meta_args(), -1, tuning)
                try:
                    mm(0)
                    torch.cuda.synchronize()
                    rec[label + "_us"] = round(graph_us(mm, max(nl, 8)), 2)
                except Exception as e:
                    rec[label + "_us"] = f"{type(e).__name__}"
            rec["layer_e2e_us"] = round(graph_us(lambda i: layers[i % nl](x), max(nl, 8)), 2)
            print(json.dumps(rec), flush=True)
        del layers
        torch.cuda.empty_cache()
