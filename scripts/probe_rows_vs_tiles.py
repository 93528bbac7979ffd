"""Round 4: where the 16-column rows kernels should hand over to the tile kernels (24 .. 64 rows) — graph-replayed `layer(x)` with the
default plan against the forced tile kernel of each family, rotating cold layers.
    python scripts/probe_rows_vs_tiles.py [proc ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
import gemlite_amd.core as C
from probe_mx_sq import graph_us  # noqa: E402

dev = torch.device("cuda:0")
H = gemlite_amd.helper
tdt = torch.float16
MS = tuple(int(v) for v in os.environ.get("GL_MS", "24,32,40,48,64").split(","))


def hqq(N, K, nbits):
    W_q = torch.randint(0, 2 ** nbits, (N, K), dtype=torch.int32, device=dev).to(torch.uint8)
    s = (torch.rand(N * K // 128, 1, device=dev) * 0.01 + 0.001).to(tdt)
    z = (torch.rand(N * K // 128, 1, device=dev) * (2 ** nbits - 1)).to(tdt)
    return W_q, s, z


def lin(N, K):
    l = torch.nn.Linear(K, N, bias=False, device=dev, dtype=tdt)
    l.weight.data /= 3.0
    return l


PROCS = {  # name -> (maker(N, K), {tag: tuning})
    "A16W4_HQQ_INT": (lambda N, K: H.A16W4_HQQ_INT(device=dev, dtype=tdt).from_weights(*hqq(N, K, 4), W_nbits=4, group_size=128),
                      {"narrow64": (0, 0, 32, 0), "narrow64x2": (0, 2, 32, 0), "mma64": (0, 0, 2, 0)}),
    "A16W8_INT8": (lambda N, K: H.A16W8(device=dev, dtype=tdt).from_weights(lin(N, K).weight.data), {"tile": (2, 0, 0, 0), "rows": (4, 0, 0, 0), "r3": (7, 0, 0, 0)}),
    "A16W8_FP8": (lambda N, K: H.A16W8_FP8(device=dev, dtype=tdt).from_weights(lin(N, K).weight.data), {"tile": (2, 0, 0, 0), "rows": (4, 0, 0, 0), "r3": (7, 0, 0, 0)}),
    "A8W8_int8_dynamic": (lambda N, K: H.A8W8_int8_dynamic(device=dev, dtype=tdt).from_weights(lin(N, K).weight.data), {"sq": (5, 0, 0, 0)}),
    "A8W4_HQQ_INT_dynamic": (lambda N, K: H.A8W4_HQQ_INT_dynamic(device=dev, dtype=tdt).from_weights(*hqq(N, K, 4)), {"mma64": (0, 0, 2, 0), "mma32": (0, 0, 1, 0)}),
    "A16W4_MXFP": (lambda N, K: H.A16W4_MXFP(device=dev, dtype=tdt).from_linear(lin(N, K), del_orig=True), {"tile": (2, 0, 0, 0), "rows": (4, 0, 0, 0), "gemv": (5, 0, 0, 0)}),
    "A16W8_MXFP": (lambda N, K: H.A16W8_MXFP(device=dev, dtype=tdt).from_linear(lin(N, K), del_orig=True), {"tile": (2, 0, 0, 0), "rows": (4, 0, 0, 0), "gemv": (5, 0, 0, 0)}),
    "A8W8_MXFP_dynamic": (lambda N, K: H.A8W8_MXFP_dynamic(device=dev, dtype=tdt).from_linear(lin(N, K), del_orig=True), {"sq": (6, 0, 0, 0)}),
    "A4W4_MXFP_dynamic": (lambda N, K: H.A4W4_MXFP_dynamic(device=dev, dtype=tdt).from_linear(lin(N, K), del_orig=True), {"sq": (6, 0, 0, 0)}),
    "A4W4_NVFP_dynamic": (lambda N, K: H.A4W4_NVFP_dynamic(device=dev, dtype=tdt).from_linear(lin(N, K), del_orig=True), {"tile": (2, 0, 0, 0)}),
}

def main():
    only = sys.argv[1:]
    for N, K in ((4096, 4096), (8192, 8192), (4096, 14336), (14336, 4096)):
        nl = max(2, min(16, (400 << 20) // (N * K)))
        for proc, (mk, alts) in PROCS.items():
            if only and proc not in only:
                continue
            layers = [mk(N, K) for _ in range(nl)]
            for M in MS:
                x = (torch.randn(M, K, device=dev) / 4).to(tdt)
                rec = dict(proc=proc, N=N, K=K, M=M)
                for tag, tun in [("default", None)] + list(alts.items()):
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


if __name__ == "__main__":
    main()
