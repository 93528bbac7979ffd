"""Round 4: graph-replayed `layer(x)` time of EVERY helper processor at 4096 x 4096 (M = 1 and 16, rotating cold layers), with the weight
bytes each launch has to move — where the decode paths stand relative to each other.
    python scripts/probe_processors.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
from gemlite_amd import _hip
import gemlite_amd.core as core

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
H = gemlite_amd.helper
N = K = 4096
tdt = torch.float16


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


def hqq(nbits):
    W_q = torch.randint(0, 2 ** nbits, (N, K), dtype=torch.int32, device=dev).to(torch.uint8)
    s = (torch.rand(N * K // 128, 1, device=dev) * 0.01 + 0.001).to(tdt)
    z = (torch.rand(N * K // 128, 1, device=dev) * (2 ** nbits - 1)).to(tdt)
    return W_q, s, z


def linear():
    lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=tdt)
    lin.weight.data /= 3.0
    return lin


MAKERS = {
    "A16W8_INT8": lambda: H.A16W8(device=dev, dtype=tdt).from_weights(linear().weight.data),
    "A16W8_FP8": lambda: H.A16W8_FP8(device=dev, dtype=tdt).from_weights(linear().weight.data),
    "A16W4_HQQ_INT": lambda: H.A16W4_HQQ_INT(device=dev, dtype=tdt).from_weights(*hqq(4), W_nbits=4, group_size=128),
    "A16W2_HQQ_INT": lambda: H.A16W2_HQQ_INT(device=dev, dtype=tdt).from_weights(*hqq(2), W_nbits=2, group_size=128),
    "A8W8_int8_dynamic": lambda: H.A8W8_int8_dynamic(device=dev, dtype=tdt).from_weights(linear().weight.data),
    "A8W8_fp8_dynamic": lambda: H.A8W8_fp8_dynamic(device=dev, dtype=tdt).from_weights(linear().weight.data),
    "A8W4_HQQ_INT_dynamic": lambda: H.A8W4_HQQ_INT_dynamic(device=dev, dtype=tdt).from_weights(*hqq(4)),
    "A8W2_HQQ_INT_dynamic": lambda: H.A8W2_HQQ_INT_dynamic(device=dev, dtype=tdt).from_weights(*hqq(2)),
    "A16W158_INT": lambda: H.A16W158_INT(device=dev, dtype=tdt).from_weights(torch.randint(-1, 2, (N, K), device=dev).to(tdt), torch.tensor(0.02)),
    "A8W158_INT_dynamic": lambda: H.A8W158_INT_dynamic(device=dev, dtype=tdt).from_weights(torch.randint(-1, 2, (N, K), device=dev).to(tdt), torch.tensor(0.02)),
    "A16W8_MXFP": lambda: H.A16W8_MXFP(device=dev, dtype=tdt).from_linear(linear(), del_orig=True),
    "A16W4_MXFP": lambda: H.A16W4_MXFP(device=dev, dtype=tdt).from_linear(linear(), del_orig=True),
    "A8W8_MXFP_dynamic": lambda: H.A8W8_MXFP_dynamic(device=dev, dtype=tdt).from_linear(linear(), del_orig=True),
    "A8W4_MXFP_dynamic": lambda: H.A8W4_MXFP_dynamic(device=dev, dtype=tdt).from_linear(linear(), del_orig=True),
    "A4W4_MXFP_dynamic": lambda: H.A4W4_MXFP_dynamic(device=dev, dtype=tdt).from_linear(linear(), del_orig=True),
    "A4W4_NVFP_dynamic": lambda: H.A4W4_NVFP_dynamic(device=dev, dtype=tdt).from_linear(linear(), del_orig=True),
}
MS = tuple(int(v) for v in os.environ.get("GL_MS", "1,16").split(","))  # GL_MS=64,256,2048: the prefill survey
only = sys.argv[1:]
for name, mk in MAKERS.items():
    if only and name not in only:
        continue
    try:
        layers = [mk() for _ in range(24)]
    except Exception as e:
        print(json.dumps(dict(proc=name, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
        continue
    wbytes = sum(t.numel() * t.element_size() for t in (layers[0].W_q, layers[0].scales, layers[0].zeros) if t is not None and t.numel() > 1)
    rec = dict(proc=name, weight_MB=round(wbytes / 1e6, 2))
    for M in MS:
        x = (torch.randn(M, K, device=dev) / 10).to(tdt)
        us = graph_us(lambda i: layers[i % 24](x), 24)
        rec[f"m{M}_us"] = round(us, 2)
        if M <= 16:
            rec[f"m{M}_frac_of_8TBs"] = round(wbytes / (us * 1e-6) / 8e12, 3)
        else:
            rec[f"m{M}_TFLOPs"] = round(2.0 * M * N * K / (us * 1e-6) / 1e12, 1)
    print(json.dumps(rec), flush=True)
    del layers
    torch.cuda.empty_cache()
