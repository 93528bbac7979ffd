"""Round 4: gemm_mx_sq_kernel (64 x 64 tiles, K unsplit) against the 128-column kernel with K slices, graph-replayed `layer(x)` (the
activation quantiser launch included in both) over rotating cold layers.
    python scripts/probe_mx_sq.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
import gemlite_amd.core as C

dev = torch.device("cuda:0")
H = gemlite_amd.helper
tdt = torch.bfloat16


def graph_us(fn, n_inner, min_seconds=0.1):
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



def main():
    for N, K in ((4096, 4096), (8192, 8192), (4096, 14336), (14336, 4096)):
        nl = max(2, min(16, (512 << 20) // (N * K)))
        for proc in ("A8W8_MXFP_dynamic", "A8W4_MXFP_dynamic", "A4W4_MXFP_dynamic"):
            layers = []
            for _ in range(nl):
                lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=tdt)
                layers.append(getattr(H, proc)(device=dev, dtype=tdt).from_linear(lin, del_orig=True))
            for M in (65, 128, 192, 256, 384, 512):
                x = (torch.randn(M, K, device=dev) / 4).to(tdt)
                rec = dict(proc=proc, N=N, K=K, M=M)
                for tag, tun in (("default", None), ("sq", (6, 0, 0, 0)), ("sq2", (6, 0, 2, 0)), ("sq3", (6, 0, 3, 0)), ("sq4", (6, 0, 4, 0)), ("mma", (2, 0, 0, 0))):
                    C.TUNING_OVERRIDE = tun
                    try:
                        rec[tag] = round(graph_us(lambda i: layers[i % nl](x), nl), 1)
                    except Exception as e:
                        rec[tag] = f"{type(e).__name__}"
                    finally:
                        C.TUNING_OVERRIDE = None
                print(json.dumps(rec), flush=True)
            del layers
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
