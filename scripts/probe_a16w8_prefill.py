import sys, os, time, json, torch
sys.path.insert(0, '/root/repo')
import gemlite_amd
H = gemlite_amd.helper
dev = torch.device("cuda:0")
def graph_us(fn, n_inner, min_seconds=0.1):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n_inner): fn(i)
        for _ in range(2): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < min_seconds:
            g.replay(); torch.cuda.synchronize(); reps += 1
        el = time.perf_counter() - t0
    torch.cuda.current_stream().wait_stream(s)
    return el / (reps * n_inner) * 1e6
N = K = 4096
for name, mk in (("A16W8_INT8", lambda: H.A16W8(device=dev, dtype=torch.float16).from_weights((torch.randn(N, K, device=dev) / 30).half())),
                 ("A16W8_FP8", lambda: H.A16W8_FP8(device=dev, dtype=torch.float16).from_weights((torch.randn(N, K, device=dev) / 30).half())),
                 ("A16W8_MXFP", lambda: H.A16W8_MXFP(device=dev, dtype=torch.float16).from_linear(torch.nn.Linear(K, N, bias=False, device=dev, dtype=torch.float16), del_orig=True)),
                 ("A16W4", lambda: H.A16W4_HQQ_INT(device=dev, dtype=torch.float16).from_weights(torch.randint(0, 16, (N, K), device=dev, dtype=torch.int32).to(torch.uint8), (torch.rand(N * K // 128, 1, device=dev) * 0.01 + 0.001).half(), (torch.rand(N * K // 128, 1, device=dev) * 15).half(), W_nbits=4, group_size=128))):
    layers = [mk() for _ in range(8)]
    rec = dict(proc=name)
    for M in (64, 128, 256, 1024, 2048):
        x = (torch.randn(M, K, device=dev) / 10).half()
        rec[f"m{M}"] = round(graph_us(lambda i: layers[i % 8](x), 8), 1)
    print(json.dumps(rec), flush=True)
