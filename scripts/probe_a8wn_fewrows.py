"""Round 4: A8Wn dynamic layers (fp8 / int8 activations x packed weights) at 2 .. 8 rows — the streaming GEMV (default up to 4 rows)
against a8wn_rows_kernel (tuning[0] = 4), graph-replayed `layer(x)` over rotating cold layers.
    python scripts/probe_a8wn_fewrows.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
import gemlite_amd.core as C
from probe_mx_sq import graph_us  # noqa: E402  (runs nothing on import? see guard below)

dev = torch.device("cuda:0")
H = gemlite_amd.helper
tdt = torch.float16


def hqq(N, K, nbits):
    W_q = torch.randint(0, 2 ** nbits, (N, K), dtype=torch.int32, device=dev).to(torch.uint8)
    s = (torch.rand(N * K // 128, 1, device=dev) * 0.01 + 0.001).to(tdt)
    z = (torch.rand(N * K // 128, 1, device=dev) * (2 ** nbits - 1)).to(tdt)
    return W_q, s, z


for N, K in ((4096, 4096), (8192, 8192), (4096, 14336), (14336, 4096)):
    nl = max(2, min(24, (300 << 20) // (N * K // 2)))
    MAKERS = {
        "A8W4_HQQ_INT_dynamic": lambda: H.A8W4_HQQ_INT_dynamic(device=dev, dtype=tdt).from_weights(*hqq(N, K, 4)),
        "A8W2_HQQ_INT_dynamic": lambda: H.A8W2_HQQ_INT_dynamic(device=dev, dtype=tdt).from_weights(*hqq(N, K, 2)),
        "A8W158_INT_dynamic": lambda: H.A8W158_INT_dynamic(device=dev, dtype=tdt).from_weights(torch.randint(-1, 2, (N, K), device=dev).to(tdt), torch.tensor(0.02)),
    }
    for proc, mk in MAKERS.items():
        layers = [mk() for _ in range(nl)]
        for M in (2, 3, 4, 5, 8):
            x = (torch.randn(M, K, device=dev) / 4).to(tdt)
            rec = dict(proc=proc, N=N, K=K, M=M)
            for tag, tun in (("default", None), ("rows", (4, 0, 0, 0))):
                C.TUNING_OVERRIDE = tun
                try:
                    rec[tag] = round(graph_us(lambda i: layers[i % nl](x), nl), 2)
                finally:
                    C.TUNING_OVERRIDE = None
            print(json.dumps(rec), flush=True)
        del layers
        torch.cuda.empty_cache()
