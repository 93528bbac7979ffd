#!/usr/bin/env python3
"""Vendor dense GEMM (torch.matmul -> hipBLASLt / rocBLAS) at the headline shapes, on the bench's clock (graph-replayed time per
launch over rotating cache-cold weights): what a dense 16-bit GEMM of the same M x N x K costs on this box.  Not a target — a
practical reference point next to the 2.5 PFLOP/s datasheet peak.   gpurun -- 'bash scripts/gpu.sh probe:probe_vendor_dense.py'"""
import json
import time

import torch

dev = torch.device("cuda:0")


def timed(fn_step, launches_per_step, min_seconds=0.2):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn_step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn_step()
    g.replay()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        g.replay()
        n += 1
        if n % 5 == 0:
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > min_seconds:
                break
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * launches_per_step) * 1e6


out = []
for (M, N, K, dt, layers) in [(256, 4096, 4096, torch.bfloat16, 16), (256, 8192, 8192, torch.bfloat16, 4), (2048, 8192, 8192, torch.bfloat16, 4),
                              (1, 4096, 4096, torch.float16, 16), (16, 4096, 4096, torch.float16, 16)]:
    g = torch.Generator(device=dev).manual_seed(0)
    Ws = [(torch.randn(N, K, generator=g, device=dev) / 30).to(dt) for _ in range(layers)]   # [N, K] like nn.Linear
    x = (torch.randn(M, K, generator=g, device=dev) / 10).to(dt)
    reps = max(1, 32 // layers)

    def step():
        for _ in range(reps):
            for W in Ws:
                torch.nn.functional.linear(x, W)

    us = timed(step, reps * layers)
    flops = 2.0 * M * N * K
    rec = {"M": M, "N": N, "K": K, "dtype": str(dt), "layers": layers, "us": round(us, 3), "tflops": round(flops / us / 1e6, 1),
           "frac_of_2500": round(flops / us / 1e6 / 2500, 3), "GBps_dense_weights": round((N * K * 2 + M * K * 2 + M * N * 2) / us / 1e3, 1)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
