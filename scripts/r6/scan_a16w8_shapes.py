"""Round 6, late: the A16W8 tile planner (int8 weights under fp16 activations; GL_W=fp8 for fp8 weights) against forced tile forms over the LLM layer
shapes of the planner fixture; layer(x) graph-replayed over rotating layers.  M values in argv (default 128 256)."""
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
FP8 = os.environ.get("GL_W", "int8") == "fp8"


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


SHAPES = [(1024, 4096), (1536, 8960), (2048, 8192), (2560, 9728), (3072, 8192), (4096, 1024), (4096, 4096), (4096, 11008), (4096, 14336), (5120, 5120), (5120, 13824), (6144, 4096),
          (8192, 2048), (8192, 3072), (8192, 8192), (8960, 1536), (11008, 4096), (12288, 4096), (13824, 5120), (14336, 4096)]
MS = [int(v) for v in sys.argv[1:]] or [128, 256]
VARIANTS = (("auto", None), ("n1", (0, 1, 32, 0)), ("n2", (0, 2, 32, 0)), ("n4", (0, 4, 32, 0)), ("w64", (0, 0, 2, 0)), ("w128", (0, 0, 4, 0)), ("w256", (0, 0, 8, 0)),
            ("w128_sk1", (0, 1, 4, 0)), ("w128_sk2", (0, 2, 4, 0)))
for (N, K) in SHAPES:
    NL = max(2, min(16, int(300e6 // (N * K))))
    mk = (lambda: H.A16W8_FP8(device=dev, dtype=tdt)) if FP8 else (lambda: H.A16W8(device=dev, dtype=tdt))
    layers = [mk().from_weights((torch.randn(N, K, device=dev) / 30).to(tdt)) for _ in range(NL)]
    for M in MS:
        x = (torch.randn(M, K, device=dev) / 10).to(tdt)
        rec = dict(M=M, N=N, K=K, us={}, kern={})
        for vn, t in VARIANTS:
            core.TUNING_OVERRIDE = t
            try:
                rec["kern"][vn] = _kernel_name(layers[0], x, -1, t if t else (0, 0, 0, 0))
                layers[0](x)
                rec["us"][vn] = round(graph_us(lambda i: layers[i % NL](x), NL), 2)
            except Exception as e:
                rec["us"][vn] = None
            finally:
                core.TUNING_OVERRIDE = None
        print(json.dumps(rec), flush=True)
    del layers
    torch.cuda.empty_cache()
