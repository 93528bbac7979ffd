"""Round 4, late: the narrow 64 x 64 tiles (KH = 4) of the 8-wave kernel for the geometries that did not have them — 8-bit activations x
packed words, block-scaled / NVFP4 / K-contiguous 8-bit weights under 16-bit activations — forced (tuning[2] = 32, K slices in
tuning[1]) against the default plan; graph-replayed `layer(x)` over rotating cold layers.
    python scripts/probe_narrow_geos.py [proc ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
import gemlite_amd.core as C
from probe_mx_sq import graph_us  # noqa: E402
from probe_rows_vs_tiles import PROCS as _P, hqq, lin, dev, tdt, H  # noqa: E402  (its sweep runs only under __main__)

MAKERS = {k: _P[k][0] for k in ("A8W4_HQQ_INT_dynamic", "A16W4_MXFP", "A16W8_MXFP", "A4W4_NVFP_dynamic", "A16W8_INT8")}
MAKERS["A8W158_INT_dynamic"] = lambda N, K: H.A8W158_INT_dynamic(device=dev, dtype=tdt).from_weights(torch.randint(-1, 2, (N, K), device=dev).to(tdt), torch.tensor(0.02))
MS = tuple(int(v) for v in os.environ.get("GL_MS", "64,128,256").split(","))
only = sys.argv[1:]
for N, K in ((4096, 4096), (8192, 8192), (14336, 4096), (8192, 2048)):
    nl = max(2, min(16, (400 << 20) // (N * K)))
    for proc, mk in MAKERS.items():
        if only and proc not in only:
            continue
        layers = [mk(N, K) for _ in range(nl)]
        for M in MS:
            x = (torch.randn(M, K, device=dev) / 4).to(tdt)
            rec = dict(proc=proc, N=N, K=K, M=M, t64=(N // 64) * ((M + 63) // 64))
            for tag, tun in (("default", None), ("n64", (0, 1, 32, 0)), ("n64x2", (0, 2, 32, 0))):
                C.TUNING_OVERRIDE = tun
                try:
                    rec[tag] = round(graph_us(lambda i: layers[i % nl](x), nl, min_seconds=0.06), 2)
                except Exception as e:
                    rec[tag] = type(e).__name__
                finally:
                    C.TUNING_OVERRIDE = None
            print(json.dumps(rec), flush=True)
        del layers
        torch.cuda.empty_cache()
