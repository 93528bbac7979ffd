"""Round 6: the x-through-LDS rows kernel of the 8-bit families (gemm_w8_rows.hip) against the register-fed kernels of rounds 3-4, one box, one process:
graph-replayed `layer(x)` over 24 rotating (HBM-cold) layers, quantiser launch of the dynamic layers included.
    python scripts/r6/probe_w8_rows_lds.py [N K]      variants: default | tuning[3] & 524288 (round-4 kernels) | tuning[0] = 4 (rows kernels past the planner's hand-over to the tiles), new and round-4"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gemlite_amd
import gemlite_amd.core as core

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
H = gemlite_amd.helper
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
tdt = torch.float16
NL = max(4, min(24, int(600e6 // (N * K))))


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


def linear():
    lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=tdt)
    lin.weight.data /= 3.0
    return lin


MAKERS = {
    "A16W8_INT8": lambda: H.A16W8(device=dev, dtype=tdt).from_weights(linear().weight.data),
    "A16W8_FP8": lambda: H.A16W8_FP8(device=dev, dtype=tdt).from_weights(linear().weight.data),
    "A8W8_int8_dynamic": lambda: H.A8W8_int8_dynamic(device=dev, dtype=tdt).from_weights(linear().weight.data),
    "A8W8_fp8_dynamic": lambda: H.A8W8_fp8_dynamic(device=dev, dtype=tdt).from_weights(linear().weight.data),
}
VARIANTS = {"default": None, "round4": (0, 0, 0, 524288), "rows_forced": (4, 0, 0, 0), "rows_forced_round4": (4, 0, 0, 524288)}  # (rows_forced: past the planner's hand-over to the tile kernels)
MS = tuple(int(v) for v in os.environ.get("GL_MS", "4,8,16,17,24,32,48,64").split(","))
for name, mk in MAKERS.items():
    layers = [mk() for _ in range(NL)]
    for M in MS:
        x = (torch.randn(M, K, device=dev) / 10).to(tdt)
        rec = dict(proc=name, N=N, K=K, M=M)
        ref = None
        for vn, t in VARIANTS.items():
            core.TUNING_OVERRIDE = t
            try:
                y = layers[0](x).float()
                torch.cuda.synchronize()
                if ref is None:
                    ref = y
                rec[vn + "_us"] = round(graph_us(lambda i: layers[i % NL](x), NL), 2)
                rec[vn + "_rel"] = float((y - ref).abs().mean() / ref.abs().mean())
            except Exception as e:
                rec[vn + "_err"] = f"{type(e).__name__}: {e}"[:160]
            finally:
                core.TUNING_OVERRIDE = None
        print(json.dumps(rec), flush=True)
    del layers
    torch.cuda.empty_cache()
