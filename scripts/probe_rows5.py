"""Round 5: the decode-shaped A16W4 rows kernel (gemm_wn_rows.hip, tuning[0] = 9) against the round-4 choice (tuning[3] & 65536) on LLM layer
shapes x M = 2 .. 64 — graph-replayed us per `layer(x)` launch over HBM-cold rotating layers (the bench's clock).  The planner's budget in
api.hip (rows5_min_m / rows5_budget_bytes) comes from this log.
    python scripts/probe_rows5.py [M ...]        GL_SHAPES="4096x4096,8192x8192" GL_DT=bf16 GL_GS=64"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
SHAPES = [(4096, 4096), (8192, 8192), (4096, 11008), (11008, 4096), (4096, 14336), (14336, 4096), (6144, 4096), (5120, 5120)]  # (N, K)
if os.environ.get("GL_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["GL_SHAPES"].split(",")]
MS = [int(a) for a in sys.argv[1:]] or [2, 4, 5, 8, 16, 24, 32, 48, 64]
DT = os.environ.get("GL_DT", "fp16")
GS = int(os.environ.get("GL_GS", "128"))
BITS = int(os.environ.get("GL_BITS", "4"))
CANDS = {"r4": (0, 0, 0, 65536), "rows5": (9, 0, 0, 0), "rows5_nt1": (9, 1, 0, 0), "rows5_nt2": (9, 2, 0, 0), "mma": (3, 0, 0, 65536), "default": None}


def time_us(mods, x, tuning, min_seconds=0.06):
    core.TUNING_OVERRIDE = tuning
    try:
        st = torch.cuda.Stream(dev)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for lin in mods[:2]:
                lin(x)
        torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        reps = max(1, -(-32 // len(mods)))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                for lin in mods:
                    lin(x)
        g.replay()
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while True:
            g.replay()
            n += 1
            if n % 5 == 0:
                torch.cuda.synchronize()
                if time.perf_counter() - t0 >= min_seconds:
                    break
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (n * reps * len(mods)) * 1e6
    finally:
        core.TUNING_OVERRIDE = None


def kname(lin, x, tuning):
    import ctypes
    a = core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
    a.matmul_type, a.M = -1, x.shape[0]
    a.x = a.out = 0x1000
    a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = x.shape[1], 1, a.N, 1
    a.input_dtype = lin.input_dtype.value
    for i in range(4):
        a.tuning[i] = (tuning or (0, 0, 0, 0))[i]
    return lib.gemlite_hip_kernel_name(ctypes.byref(a)).decode()


for (N, K) in SHAPES:
    name = f"probe_{N}x{K}"
    nl = max(2, min(32, int(300e6 // (N * K * BITS // 8))))
    bench.WORKLOADS[name] = (N, K, BITS, GS, 1, DT, nl, "hbm")
    mods, _ = bench.build_layers(name, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    for M in MS:
        x = (torch.randn(M, K, generator=g, device=dev) / 10).to(torch.float16 if DT == "fp16" else torch.bfloat16)
        res, names = {}, {}
        for label, t in CANDS.items():
            kn = kname(mods[0], x, t)
            if label.startswith("rows5") and "_rows_kernel" not in kn:
                continue
            try:
                res[label] = round(time_us(mods, x, t), 2)
                names[label] = kn
                if label == "mma":
                    res["mma_kernel"] = kn
            except Exception as e:  # noqa: BLE001
                res[label] = None
                names[label] = str(e)[:60]
        print(json.dumps(dict(N=N, K=K, M=M, gs=GS, dt=DT, us=res, r4_kernel=names.get("r4"), default_kernel=names.get("default"),
                              x_reread_MB=round(N / 16 * ((M + 15) // 16 * 16) * K * 2 / 2**20, 1))), flush=True)
    del mods
    torch.cuda.empty_cache()
