#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (contract in the task prompt, §④).

Default workload (BASELINE.json configs[1]): A16W4 group_size=128, 4096x4096, M=1 decode GEMV.
A "step" is one pass of the hot path over a stack of LAYERS (=32) DISTINCT packed layers with the same
input row — one decode step through 32 linear layers.  32 x 8.93 MB = 286 MB of weights exceeds the 256 MiB
Infinity Cache, so every layer's stream comes from HBM (cache-cold rotation, SURVEY.md §7 "Hard parts").
The step is captured once as a hipGraph (the launch-bound regime the reference itself addresses with CUDA
graphs, config.py:17) and replayed; `value` is whole-job algorithmic GB/s over all ranks.

  roofline     — dominant kernel's ALGORITHMIC bytes (or flops) per launch / its device duration.  The
                 duration is measured live with HIP events attached to individual launches
                 (hipExtLaunchKernel start/stop events through gemlite_hip_set_profile_events) in an eager
                 pass over the same rotating layers right after the timed region; `gap_inclusive` repeats
                 the figure with the timed region's wall time / launches (kernel + launch gaps).
  cpu_baseline — oracle/torch_cpu_path.py (a port of the reference's test oracle, all host cores), rank 0,
                 N=1 only, bounded to ~12 s.

Other workloads (for development / profiles): --workload a16w4_4096_m256 | a16w4_8192_m256 | a16w2_16384_m1 ...
Multi-GPU: the path does not shard (SURVEY.md §8 e) -> N independent replicas, no collective in the data path.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak

WORKLOADS = {
    # name: (N, K, nbits, group, M, dtype, layers, bound)
    "a16w4_4096_m1": (4096, 4096, 4, 128, 1, "fp16", 32, "hbm"),
    "a16w4_4096_m1_bf16": (4096, 4096, 4, 128, 1, "bf16", 32, "hbm"),
    "a16w4_4096_m8": (4096, 4096, 4, 128, 8, "fp16", 32, "hbm"),
    "a16w4_4096_m16": (4096, 4096, 4, 128, 16, "fp16", 32, "hbm"),
    "a16w4_4096_m256": (4096, 4096, 4, 128, 256, "bf16", 32, "mfma"),
    "a16w4_8192_m256": (8192, 8192, 4, 128, 256, "bf16", 8, "mfma"),
    "a16w4_8192_m1": (8192, 8192, 4, 128, 1, "fp16", 8, "hbm"),
    "a16w2_16384_m1": (16384, 16384, 2, 128, 1, "fp16", 4, "hbm"),
    "a16w4_16384_m1": (16384, 16384, 4, 128, 1, "fp16", 2, "hbm"),
    # BASELINE config 4: A8W8 int8 dynamic (x pre-quantised per token outside the timed matmul; group = K: channel-wise)
    "a8w8_4096_m1": (4096, 4096, 8, 4096, 1, "int8", 16, "hbm"),
    "a8w8_4096_m16": (4096, 4096, 8, 4096, 16, "int8", 16, "hbm"),
    "a8w8_4096_m256": (4096, 4096, 8, 4096, 256, "int8", 16, "mfma"),
}
INT8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA peak (2x bf16, MI355X_MICROARCH.md)


def algorithmic_bytes(M, N, K, nbits, group, esize=2):
    """K*N*b/8 + 2*(K/g)*N*sizeof(meta) + M*K*sizeof(x) + M*N*sizeof(out)."""
    return K * N * nbits // 8 + 2 * (K // group) * N * esize + M * K * esize + M * N * esize


class HipEvents:
    """Minimal hipEvent access through libamdhip64 (the runtime torch already loaded)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def create(self):
        e = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def elapsed_ms(self, a, b):
        self.hip.hipEventSynchronize(b)
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b)
        return ms.value if rc == 0 else float("nan")


def build_layers(name, device):
    import gemlite_amd
    from gemlite_amd import GemLiteLinear
    from gemlite_amd.dtypes import TORCH_TO_DTYPE

    N, K, nbits, group, M, dt, layers, bound = WORKLOADS[name]
    if dt == "int8":
        from gemlite_amd.helper import A8W8_int8_dynamic
        from gemlite_amd.quant_utils import scale_activations_per_token
        g = torch.Generator(device="cpu").manual_seed(0)
        proc = A8W8_int8_dynamic(device=device, dtype=torch.float16)
        mods = [proc.from_weights((torch.randn(N, K, generator=g) / 30).half()) for _ in range(layers)]
        x = (torch.randn(M, K, generator=g) / 10).half().to(device)
        return mods, scale_activations_per_token(x, torch.int8)  # (x_q int8 [M, K], scales_x fp32 [M, 1])
    tdt = torch.float16 if dt == "fp16" else torch.bfloat16
    code = TORCH_TO_DTYPE[tdt]
    g = torch.Generator(device="cpu").manual_seed(0)
    mods = []
    for _ in range(layers):
        W_q = torch.randint(0, 2 ** nbits, (N, K), generator=g, dtype=torch.int32).to(torch.uint8).to(device)
        scales = (torch.rand(N * K // group, 1, generator=g) * 0.01 + 0.001).to(tdt).to(device)
        zeros = (torch.rand(N * K // group, 1, generator=g) * (2 ** nbits - 1)).to(tdt).to(device)
        lin = GemLiteLinear(nbits, group, K, N, code, code)
        lin.pack(W_q, scales, zeros, None)
        mods.append(lin)
        del W_q
    x = (torch.randn(M, K, generator=g) / 10).to(tdt).to(device)  # random, not zeros (DVFS give-back)
    return mods, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="a16w4_4096_m1", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-samples", type=int, default=256, help="launches timed individually for roofline")
    ap.add_argument("--tuning", default="", help="development: comma-separated tuning[] override, e.g. 4,8")
    ap.add_argument("--matmul-type", default="", help="development: force a kernel family (forward_manual)")
    args = ap.parse_args()

    from gemlite_amd.bench_utils import ReplicaGroup, timed_steps, whole_job_rate
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rgroup = ReplicaGroup("nccl")  # RCCL; used only for the timing barrier / max-reduce (replicas, no data exchange)
    world, rank = rgroup.world, rgroup.rank

    from gemlite_amd import _hip
    lib = _hip.load()  # fails loudly if the HIP library is missing
    if args.tuning:
        import gemlite_amd.core as _core
        t = [int(v) for v in args.tuning.split(",")]
        _core.TUNING_OVERRIDE = tuple(t + [0] * (4 - len(t)))

    N, K, nbits, group, M, dt, layers, bound = WORKLOADS[args.workload]
    mods, x = build_layers(args.workload, device)

    def call(lin):
        if dt == "int8":  # the matmul alone: x was quantised once in build_layers
            from gemlite_amd.core import _hip_matmul
            return _hip_matmul(x[0], lin.W_q, lin.scales, lin.zeros, x[1], lin.get_meta_args(), -1)
        return lin.forward_manual(x, args.matmul_type) if args.matmul_type else lin(x)

    def step_eager():
        for lin in mods:
            call(lin)

    # warm-up on a side stream (allocates the per-stream split-K workspace), then capture one step
    stream = torch.cuda.Stream(device)
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(2):
            step_eager()
    torch.cuda.current_stream().wait_stream(stream)
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            step_eager()

    def run_step():
        if graph is not None:
            graph.replay()
        else:
            with torch.cuda.stream(stream):
                step_eager()

    elapsed = timed_steps(rgroup, run_step, args.steps, args.warmup, device_sync=torch.cuda.synchronize, device=device)

    launches = args.steps * layers
    bytes_per_launch = algorithmic_bytes(M, N, K, nbits, group)
    if dt == "int8":  # int8 W + fp32 channel scales + int8 x + fp32 token scales + fp16 out
        bytes_per_launch = K * N + N * 4 + M * K + M * 4 + M * N * 2
    flops_per_launch = 2 * M * N * K
    ms_per_step = elapsed / args.steps * 1e3

    # ---- per-kernel device duration: HIP events attached to individual launches (eager, same rotation) ----
    kernel_us, kernel_name = float("nan"), "?"
    try:
        ev = HipEvents()
        pairs = [(ev.create(), ev.create()) for _ in range(min(args.kernel_samples, 1024))]
        with torch.cuda.stream(stream):
            for i, (a, b) in enumerate(pairs):
                lib.gemlite_hip_set_profile_events(a, b)
                call(mods[i % layers])
        torch.cuda.synchronize()
        durs = np.array([ev.elapsed_ms(a, b) * 1e3 for a, b in pairs])
        durs = durs[np.isfinite(durs) & (durs > 0)]
        if durs.size:
            kernel_us = float(durs.mean())
        from gemlite_amd.core import _static_args
        a0 = _static_args(mods[0].W_q, mods[0].scales, mods[0].zeros, mods[0].get_meta_args())
        kernel_name = lib.gemlite_hip_kernel_name(ctypes.byref(a0)).decode()  # a0 still holds the last launch's args
    except Exception as e:  # keep the bench line even if the event path is unavailable
        print(f"[bench] per-kernel event timing unavailable: {e}", file=sys.stderr)

    gap_us = elapsed / launches * 1e6
    if bound == "hbm":
        unit, peak = "GB/s", HBM_PEAK_GBS
        value = whole_job_rate(layers * bytes_per_launch, args.steps, world, elapsed) / 1e9
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9 if kernel_us == kernel_us else bytes_per_launch / (gap_us * 1e-6) / 1e9
        gap_incl = bytes_per_launch / (gap_us * 1e-6) / 1e9
        metric = "HBM GB/s (algorithmic bytes) vs roofline, A16W4 gs=128 4096x4096 M=1"
    else:
        unit, peak = "TFLOP/s", (INT8_MFMA_PEAK_TOPS if dt == "int8" else MFMA_PEAK_TFLOPS)
        value = whole_job_rate(layers * flops_per_launch, args.steps, world, elapsed) / 1e12
        achieved = flops_per_launch / (kernel_us * 1e-6) / 1e12 if kernel_us == kernel_us else flops_per_launch / (gap_us * 1e-6) / 1e12
        gap_incl = flops_per_launch / (gap_us * 1e-6) / 1e12
        metric = "TFLOP/s vs bf16 MFMA roofline, A16W4 gs=128 M=256"

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload)
        except Exception:
            traffic = None

    line = {
        "metric": metric if args.workload == "a16w4_4096_m1" else f"{unit} {args.workload}",
        "value": round(value, 3), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dt, "data": "synthetic (seeded random W_q/scales/zeros/x, random-init)",
        "config": {"workload": f"A{8 if dt == 'int8' else 16}W{nbits} gs={group} {N}x{K} M={M} {dt}; step = {layers} distinct layers "
                               f"(cache-cold rotation), {'hipGraph replay' if graph is not None else 'eager'}",
                   "layers_per_step": layers, "launches_per_step": layers, "parallelism": f"replicas x{world}",
                   **({"tuning": args.tuning} if args.tuning else {}),
                   **({"matmul_type": args.matmul_type} if args.matmul_type else {})},
        "roofline": {"bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                     "frac": round(achieved / peak, 4), "traffic": traffic, "kernel": kernel_name,
                     "kernel_us": None if kernel_us != kernel_us else round(kernel_us, 3),
                     "algorithmic_bytes_per_launch": bytes_per_launch, "flops_per_launch": flops_per_launch,
                     "gap_inclusive": round(gap_incl, 3), "us_per_launch_in_timed_region": round(gap_us, 3),
                     "frac_vs_measured_copy_6290": round(achieved / 6290.0, 4) if bound == "hbm" else None},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.torch_cpu_path import time_cpu_baseline
        sec, calls, threads = time_cpu_baseline(M, N, K, nbits, group, budget_s=12.0)
        cpu_val = (bytes_per_launch / sec / 1e9) if bound == "hbm" else (flops_per_launch / sec / 1e12)
        line["cpu_baseline"] = {"value": round(cpu_val, 5), "unit": unit, "cores": threads, "kind": "port",
                                "sample": f"{calls} calls of unpack+dequant+matmul (torch CPU, fp32) on one {N}x{K} layer, "
                                          f"M={M}, {sec * 1e3:.2f} ms/call"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    rgroup.close()


if __name__ == "__main__":
    main()
