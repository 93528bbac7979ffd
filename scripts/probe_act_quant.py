"""Per-token activation quantiser (round 3: row in registers) and the whole dynamic-quant forward `layer(x)` of the A8W8 processors:
graph-replayed time per call.    python scripts/probe_act_quant.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
from gemlite_amd import helper as H
from gemlite_amd.quant_utils import scale_activations_per_token

dev = "cuda:0"
torch.cuda.set_device(0)


def graph_us(fn, reps=32, min_seconds=0.15):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    e0.record()
    while True:
        for _ in range(10):
            g.replay()
        n += 10
        e1.record(); e1.synchronize()
        if e0.elapsed_time(e1) > min_seconds * 1e3:
            break
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


for qdt in (torch.int8, torch.float8_e4m3fn):
    for (M, K) in ((1, 4096), (16, 4096), (64, 4096), (256, 4096), (64, 8192), (256, 16384), (16, 11008)):
        for tdt in (torch.float16, torch.bfloat16):
            x = (torch.randn(M, K, device=dev) / 10).to(tdt)
            us = graph_us(lambda: scale_activations_per_token(x, qdt))
            print(json.dumps(dict(what="act_quant_per_token", out=str(qdt)[6:], M=M, K=K, x=str(tdt)[6:], graph_us=round(us, 3))), flush=True)
for kind in ("int8", "fp8"):
    torch.manual_seed(0)
    W = (torch.randn(4096, 4096) / 30).half()
    proc = H.A8W8_int8_dynamic(device=dev, dtype=torch.float16) if kind == "int8" else H.A8W8_dynamic(device=dev, dtype=torch.float16, fp8=torch.float8_e4m3fn)
    lin = proc.from_weights(W)
    for M in (1, 16, 32, 64, 256):
        x = (torch.randn(M, 4096, device=dev) / 10).half()
        us = graph_us(lambda: lin(x))
        print(json.dumps(dict(what="layer(x) = quantiser + matmul", kind=kind, M=M, N=4096, K=4096, graph_us=round(us, 3))), flush=True)
