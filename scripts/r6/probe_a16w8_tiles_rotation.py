"""Round 6: K rotation on the narrow tiles of the 8-bit weight-only tile kernel (gemm_a16w8_kernel<64x64>, gemm_wn_mma_kernel.inc with the K-contiguous 8-bit
geometry) — experiment switch tuning[3] & 16777216 against & 4194304 (none); layer(x) over rotating layers."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gemlite_amd
import gemlite_amd.core as core
from tests.test_gpu_parity import _kernel_name

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
H = gemlite_amd.helper
tdt = torch.float16


def graph_us(fn, n_inner, min_seconds=0.2):
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


for (N, K) in ((4096, 4096), (2048, 8192), (8192, 2048)):
    NL = max(4, min(24, int(600e6 // (N * K))))
    layers = [H.A16W8(device=dev, dtype=tdt).from_weights((torch.randn(N, K, device=dev) / 30).to(tdt)) for _ in range(NL)]
    for M in (128, 192, 256):
        x = (torch.randn(M, K, device=dev) / 10).to(tdt)
        rec = dict(N=N, K=K, M=M)
        for rep in range(2):
            for vn, t in (("none", (0, 0, 0, 4194304)), ("rot", (0, 0, 0, 16777216))):
                core.TUNING_OVERRIDE = t
                try:
                    rec["kernel"] = _kernel_name(layers[0], x, -1, t)
                    rec[f"{vn}{rep}_us"] = round(graph_us(lambda i: layers[i % NL](x), NL), 2)
                finally:
                    core.TUNING_OVERRIDE = None
        print(json.dumps(rec), flush=True)
    del layers
    torch.cuda.empty_cache()
