#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (contract in the task prompt, §④).

Default workload (BASELINE.json configs[1]): A16W4 group_size=128, 4096x4096, M=1 decode GEMV.
A "step" is one pass of the hot path over a stack of LAYERS (=32) DISTINCT packed layers with the same
input row — one decode step through 64 linear layers.  64 x 8.93 MB = 572 MB of weights is more than twice the 256 MiB
Infinity Cache (SURVEY.md §8(d) asks for >= 512 MiB; rounds 1-3 rotated 286 MB — `rotation_ab` in the line times both), so every
layer's stream comes from HBM (cache-cold rotation, SURVEY.md §7 "Hard parts").
The step is captured once as a hipGraph (the launch-bound regime the reference itself addresses with CUDA
graphs, config.py:17) and replayed; `value` is whole-job algorithmic GB/s over all ranks.

The JSON line carries BOTH halves of BASELINE.json's metric ("... at M=1 and M=256") and all five BASELINE configs, measured in this
process.  Round 5: EVERY block below lives INSIDE the `roofline` object in compact form ({kernel, kernel_us, frac, traffic, mfma_util};
`achieved` / `unit` per block: --full-out) — the driver's parsed record keeps `roofline` whole and only the NAMES of other top-level keys, so the M = 256
half of the metric (`roofline.m256`) is now driver-retained evidence.  `--full-out PATH` writes the verbose blocks to a file.  Every block uses ONE clock: wall time of a replayed hipGraph that holds >= 32 back-to-back launches of the workload over
rotating (cache-cold) layers, divided by the launches — the same quantity as the timed region of `value`, and the one that
reproduces from `rocprofv3 --kernel-trace --stats` of this command (profiles/r03/official/: the kernel's average duration agrees
within a few percent; the graph's dependent-launch boundary, ~1.5 us, is inside it).  Per-launch HIP events are reported as a
secondary figure only (`event_us`; an EMPTY kernel reads ~4 us through them: `event_clock_floor_us`).
  roofline        — M=1 (the `value` workload): algorithmic bytes per launch / time per launch of the TIMED REGION.
  roofline.m256   — cfgA (4096^2) and cfgB (8192^2, BASELINE configs[2]) at M=256 in bf16: TFLOP/s against the dense bf16 MFMA peak;
                    `mfma_util` = matrix-pipe busy share from the committed SQ counter passes (profiles/mfma_util.json), or null.
  roofline.cfg4   — BASELINE configs[3]: A8W8 int8 4096^2 at M = 1 / 16 / 256 (+ 32 / 64, round 6): the matmul alone (x pre-quantised outside the timed
                    region) AND `*_layer_e2e`: layer(x) with the dynamic activation quantisation inside the timed region (round 4).
  roofline.cfg5   — BASELINE configs[4]: A16W2 g128 and FP8 x FP8 16384^2 at M = 1 / 256 (+ `*_layer_e2e` for FP8).
  roofline.m1_bf16, roofline.rotation_ab — the bf16 twin of the headline; the headline step over 32 (286 MB) vs 64 (572 MB) distinct layers.
  roofline.mx_fewrows, roofline.mx_m256 — block-scaled formats (MXFP8 / MXFP4 / NVFP4) at 4096^2: M = 16 (few-row kernels) and M = 256
                    (unsplit 64 x 64 scaled-MFMA tiles, round 4), matmul alone and `*_layer_e2e` with the activation quantiser inside.
  roofline.prefill_m2048 — 8192^2 at M=2048 (bf16), the large-M end of the MFMA kernel family.
  roofline.trend_m1 — the same GEMV family at 8192^2 and 16384^2 (fraction of HBM peak grows with size).
  roofline.fewrows — A16W4 4096^2 at M = 16 / 32 / 64 in fp16 (decode-batch sizes; round 5: the 16-column MFMA rows kernel).
  roofline.sustained — >= 6 s of back-to-back replays of the headline step (an outside sampler sees the GPU busy).
  cpu_baseline    — oracle/torch_cpu_path.py (a port of the reference's test oracle: unpack -> dequant -> matmul in
                    torch CPU ops; thread count swept, best reported, plus the matmul-only variant), rank 0, N=1 only.

Other workloads (development / profiles): --workload a16w4_4096_m256 | a16w4_8192_m256 | a16w2_16384_m1 ... [--single]
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
INT8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA peak (2x bf16, MI355X_MICROARCH.md)
MXFP8_MFMA_PEAK_TFLOPS = 5000.0   # block-scaled fp8 (v_mfma_scale_*_f8f6f4), dense
MXFP4_MFMA_PEAK_TFLOPS = 10000.0  # block-scaled fp4 / fp6, dense

WORKLOADS = {
    # name: (N, K, nbits, group, M, dtype, layers, bound)
    "a16w4_4096_m1": (4096, 4096, 4, 128, 1, "fp16", 64, "hbm"),
    "a16w4_4096_m1_bf16": (4096, 4096, 4, 128, 1, "bf16", 64, "hbm"),
    "a16w4_4096_m8": (4096, 4096, 4, 128, 8, "fp16", 32, "hbm"),
    "a16w4_4096_m16": (4096, 4096, 4, 128, 16, "fp16", 32, "hbm"),
    "a16w4_4096_m32": (4096, 4096, 4, 128, 32, "fp16", 32, "hbm"),
    "a16w4_4096_m64": (4096, 4096, 4, 128, 64, "fp16", 32, "hbm"),
    "a16w4_4096_m256": (4096, 4096, 4, 128, 256, "bf16", 64, "mfma"),
    "a16w4_4096_m256_fp16": (4096, 4096, 4, 128, 256, "fp16", 32, "mfma"),
    "a16w4_8192_m256": (8192, 8192, 4, 128, 256, "bf16", 16, "mfma"),
    "a16w4_8192_m2048": (8192, 8192, 4, 128, 2048, "bf16", 8, "mfma"),   # prefill-sized M: the tiles fill the chip without K slices
    "a16w4_4096_m2048": (4096, 4096, 4, 128, 2048, "bf16", 32, "mfma"),
    "a16w4_8192_m8192": (8192, 8192, 4, 128, 8192, "bf16", 8, "mfma"),
    "a16w4_8192_m1": (8192, 8192, 4, 128, 1, "fp16", 16, "hbm"),
    "a16w2_16384_m1": (16384, 16384, 2, 128, 1, "fp16", 8, "hbm"),
    "a16w2_16384_m256": (16384, 16384, 2, 128, 256, "bf16", 8, "mfma"),
    "a16w4_16384_m1": (16384, 16384, 4, 128, 1, "fp16", 4, "hbm"),
    "a16w4_11008_m1": (4096, 11008, 4, 128, 1, "fp16", 12, "hbm"),
    # BASELINE config 4: A8W8 int8 dynamic (x pre-quantised per token outside the timed matmul; group = K: channel-wise)
    "a8w8_4096_m1": (4096, 4096, 8, 4096, 1, "int8", 32, "hbm"),
    "a8w8_4096_m16": (4096, 4096, 8, 4096, 16, "int8", 32, "hbm"),
    "a8w8_4096_m32": (4096, 4096, 8, 4096, 32, "int8", 32, "hbm"),   # (round 6: the decode batches between the named 16 and 256 — VERDICT r5 item 5)
    "a8w8_4096_m64": (4096, 4096, 8, 4096, 64, "int8", 32, "hbm"),
    "a8w8_4096_m256": (4096, 4096, 8, 4096, 256, "int8", 32, "mfma"),
    # BASELINE config 5, second half: FP8 x FP8 (e4m3, per-token x per-channel scales), 16384 x 16384
    "fp8_16384_m1": (16384, 16384, 8, 16384, 1, "fp8w8", 2, "hbm"),
    "fp8_16384_m256": (16384, 16384, 8, 16384, 256, "fp8w8", 2, "mfma"),
    # A8Wn dynamic (helper.py:502-615): fp8 e4m3 activations (pre-quantised per token) x 4-bit g128 weights, fp16 out
    "a8w4_4096_m1": (4096, 4096, 4, 128, 1, "fp8", 32, "hbm"),
    "a8w4_4096_m16": (4096, 4096, 4, 128, 16, "fp8", 32, "hbm"),
    "a8w4_4096_m256": (4096, 4096, 4, 128, 256, "fp8", 32, "mfma"),
    "a8w4_8192_m256": (8192, 8192, 4, 128, 256, "fp8", 8, "mfma"),
    # block-scaled formats (helper.py:372-400, 658-950): group = 32, weights quantised by WeightQuantizerMXFP.  dtype names the
    # layer: mxa8 = MXFP8 activations with e8m0 microscales (pre-quantised outside the timed matmul), mxa4 = MXFP4 activations,
    # mxa16 = bf16 activations x MX weights
    "mx_a8w8_4096_m1": (4096, 4096, 8, 32, 1, "mxa8", 16, "hbm"),
    "mx_a8w8_4096_m256": (4096, 4096, 8, 32, 256, "mxa8", 16, "mfma"),
    "mx_a8w8_8192_m256": (8192, 8192, 8, 32, 256, "mxa8", 4, "mfma"),
    "mx_a8w8_8192_m2048": (8192, 8192, 8, 32, 2048, "mxa8", 4, "mfma"),
    "mx_a8w4_8192_m256": (8192, 8192, 4, 32, 256, "mxa8", 8, "mfma"),
    "mx_a8w4_8192_m2048": (8192, 8192, 4, 32, 2048, "mxa8", 8, "mfma"),
    "mx_a4w4_4096_m1": (4096, 4096, 4, 32, 1, "mxa4", 32, "hbm"),
    "mx_a4w4_8192_m256": (8192, 8192, 4, 32, 256, "mxa4", 8, "mfma"),
    "mx_a4w4_4096_m256": (4096, 4096, 4, 32, 256, "mxa4", 32, "mfma"),
    "mx_a4w4_8192_m2048": (8192, 8192, 4, 32, 2048, "mxa4", 8, "mfma"),
    # round 4: decode-batch sizes of the block-scaled formats (few-row scaled-MFMA kernel) and NVFP4 (fp16 MFMA tile kernel)
    "mx_a8w8_4096_m16": (4096, 4096, 8, 32, 16, "mxa8", 32, "hbm"),
    "mx_a4w4_4096_m16": (4096, 4096, 4, 32, 16, "mxa4", 32, "hbm"),
    "nvfp4_4096_m16": (4096, 4096, 4, 16, 16, "nva4", 32, "hbm"),
    "nvfp4_4096_m256": (4096, 4096, 4, 16, 256, "nva4", 32, "mfma"),
    "mx_a16w4_4096_m1": (4096, 4096, 4, 32, 1, "mxa16", 32, "hbm"),
    "mx_a16w4_4096_m256": (4096, 4096, 4, 32, 256, "mxa16", 32, "mfma"),
    "mx_a16w4_8192_m256": (8192, 8192, 4, 32, 256, "mxa16", 8, "mfma"),
    "mx_a16w8_8192_m256": (8192, 8192, 8, 32, 256, "mxa16", 4, "mfma"),
}
PREQUANT = ("int8", "fp8", "fp8w8", "mxa8", "mxa4", "nva4")  # workloads whose x is quantised once, outside the timed matmul
MX = ("mxa8", "mxa4", "mxa16", "nva4")


def algorithmic_bytes(M, N, K, nbits, group, esize=2):
    """K*N*b/8 + 2*(K/g)*N*sizeof(meta) + M*K*sizeof(x) + M*N*sizeof(out)."""
    return K * N * nbits // 8 + 2 * (K // group) * N * esize + M * K * esize + M * N * esize


def work_per_launch(name):
    N, K, nbits, group, M, dt, layers, bound = WORKLOADS[name]
    nbytes = algorithmic_bytes(M, N, K, nbits, group)
    if dt in ("int8", "fp8w8"):  # 8-bit W + fp32 channel scales + 8-bit x + fp32 token scales + fp16 out
        nbytes = K * N + N * 4 + M * K + M * 4 + M * N * 2
    if dt == "fp8":  # packed W + fp16 group metadata + fp8 x + fp32 token scales + fp16 out
        nbytes = K * N * nbits // 8 + 2 * (K // group) * N * 2 + M * K + M * 4 + M * N * 2
    if dt in MX:  # W elements + one scale byte per 32 k, x elements (+ its scale bytes), bf16 out
        xb = {"mxa8": M * K + M * K // 32, "mxa4": M * K // 2 + M * K // 32, "mxa16": M * K * 2, "nva4": M * K // 2 + M * K // 16}[dt]
        nbytes = K * N * nbits // 8 + (K // 32) * N + xb + M * N * 2
    return nbytes, 2 * M * N * K


def build_layers(name, device, layers=None, raw_x=False):
    from gemlite_amd import GemLiteLinear
    from gemlite_amd.dtypes import TORCH_TO_DTYPE

    N, K, nbits, group, M, dt, nl, bound = WORKLOADS[name]
    layers = nl if layers is None else layers
    if dt in MX:
        from gemlite_amd import helper as H
        from gemlite_amd.quant_utils import scale_activations_mxfp4, scale_activations_mxfp8, scale_activations_nvfp4
        g = torch.Generator(device=device).manual_seed(0)
        tdt = torch.bfloat16
        proc = {"mxa8": lambda: H.A8Wn_MXFP_dynamic(device=device, dtype=tdt, post_scale=False, W_nbits=nbits),
                "mxa4": lambda: H.A4W4_MXFP_dynamic(device=device, dtype=tdt),
                "nva4": lambda: H.A4W4_NVFP_dynamic(device=device, dtype=tdt),
                "mxa16": lambda: H.A16Wn_MXFP(device=device, dtype=tdt, W_nbits=nbits)}[dt]()
        mods = []
        for _ in range(layers):
            lin = torch.nn.Linear(K, N, bias=False, device=device, dtype=tdt)
            with torch.no_grad():
                lin.weight.copy_((torch.randn(N, K, generator=g, device=device) / 30).to(tdt))
            mods.append(proc.from_linear(lin, del_orig=True))
        x = (torch.randn(M, K, generator=g, device=device) / 10).to(tdt)
        if dt == "mxa16":
            return mods, x
        if raw_x:
            return mods, x
        return mods, (scale_activations_mxfp8(x) if dt == "mxa8" else (scale_activations_nvfp4(x) if dt == "nva4" else scale_activations_mxfp4(x)))
    if dt in ("int8", "fp8w8"):
        from gemlite_amd.helper import A8W8_fp8_dynamic, A8W8_int8_dynamic
        from gemlite_amd.quant_utils import scale_activations_per_token
        g = torch.Generator(device=device).manual_seed(0)
        proc = (A8W8_int8_dynamic if dt == "int8" else A8W8_fp8_dynamic)(device=device, dtype=torch.float16)
        mods = [proc.from_weights((torch.randn(N, K, generator=g, device=device) / 30).half()) for _ in range(layers)]
        x = (torch.randn(M, K, generator=g, device=device) / 10).half()
        if raw_x:
            return mods, x
        return mods, scale_activations_per_token(x, torch.int8 if dt == "int8" else torch.float8_e4m3fn)  # (x_q [M, K], scales_x fp32 [M, 1])
    if dt == "fp8":
        from gemlite_amd.helper import A8Wn_HQQ_INT_dynamic
        from gemlite_amd.quant_utils import scale_activations_per_token
        g = torch.Generator(device=device).manual_seed(0)
        proc = A8Wn_HQQ_INT_dynamic(device=device, dtype=torch.float16, W_nbits=nbits)
        mods = []
        for _ in range(layers):
            W_q = torch.randint(0, 2 ** nbits, (N, K), generator=g, dtype=torch.int32, device=device).to(torch.uint8)
            scales = (torch.rand(N * K // group, 1, generator=g, device=device) * 0.01 + 0.001).half()
            zeros = (torch.rand(N * K // group, 1, generator=g, device=device) * (2 ** nbits - 1)).half()
            mods.append(proc.from_weights(W_q, scales, zeros))
        x = (torch.randn(M, K, generator=g, device=device) / 10).half()
        return mods, scale_activations_per_token(x, torch.float8_e4m3fn)
    tdt = torch.float16 if dt == "fp16" else torch.bfloat16
    code = TORCH_TO_DTYPE[tdt]
    g = torch.Generator(device=device).manual_seed(0)  # seeded device RNG: the 16384^2 layers would take seconds on the host
    mods = []
    for _ in range(layers):
        W_q = torch.randint(0, 2 ** nbits, (N, K), generator=g, dtype=torch.int32, device=device).to(torch.uint8)
        scales = (torch.rand(N * K // group, 1, generator=g, device=device) * 0.01 + 0.001).to(tdt)
        zeros = (torch.rand(N * K // group, 1, generator=g, device=device) * (2 ** nbits - 1)).to(tdt)
        lin = GemLiteLinear(nbits, group, K, N, code, code)
        lin.pack(W_q, scales, zeros, None)
        mods.append(lin)
        del W_q
    x = (torch.randn(M, K, generator=g, device=device) / 10).to(tdt)  # random, not zeros (DVFS give-back)
    return mods, x


class Runner:
    """One workload: eager step, captured hipGraph of the step, per-launch event timing.
    e2e (dynamic-quantisation workloads): time `layer(x)` on the UNQUANTISED 16-bit x — the product path, quantiser included (one fused
    launch at M = 1, quantiser + matmul above) — instead of the matmul alone on a pre-quantised x."""

    def __init__(self, name, device, lib, layers=None, matmul_type="", use_graph=True, e2e=False):
        self.name, self.device, self.lib, self.matmul_type = name, device, lib, matmul_type
        self.e2e = e2e
        self.N, self.K, self.nbits, self.group, self.M, self.dt, _, self.bound = WORKLOADS[name]
        self.mods, self.x = build_layers(name, device, layers, raw_x=e2e)
        self.layers = len(self.mods)
        self.bytes, self.flops = work_per_launch(name)
        self.stream = torch.cuda.Stream(device)
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):  # warm-up on the side stream (allocates the per-stream split-K workspace)
            for _ in range(2):
                self.step_eager()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = self.chain_graph = None
        self.chain_reps = 1
        if use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.step_eager()
            # chained timing: one graph replay costs the host / command processor ~10 us whatever it holds (guide row
            # graph-replay-floor), so short steps (2 .. 8 launches) are repeated inside ONE graph until it holds >= 32 launches
            self.chain_reps = max(1, -(-32 // self.layers))
            if self.chain_reps > 1:
                self.chain_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.chain_graph, stream=self.stream):
                    for _ in range(self.chain_reps):
                        self.step_eager()
            else:
                self.chain_graph = self.graph

    def call(self, lin):
        if self.e2e:  # the layer as the user calls it: dynamic activation quantisation included
            return lin(self.x)
        if self.dt in PREQUANT:  # the matmul alone: x was quantised once in build_layers
            from gemlite_amd.core import _hip_matmul
            return _hip_matmul(self.x[0], lin.W_q, lin.scales, lin.zeros, self.x[1], lin.get_meta_args(), -1)
        return lin.forward_manual(self.x, self.matmul_type) if self.matmul_type else lin(self.x)

    def step_eager(self):
        for lin in self.mods:
            self.call(lin)

    def run_step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            with torch.cuda.stream(self.stream):
                self.step_eager()

    def run_chain(self):
        if self.chain_graph is not None:
            self.chain_graph.replay()
        else:
            with torch.cuda.stream(self.stream):
                self.step_eager()

    def chained_us_per_launch(self, min_seconds=0.05, min_steps=5):
        """Wall time per launch of back-to-back steps (kernel + the dependent-launch gap), un-profiled."""
        reps = self.chain_reps if self.chain_graph is not None else 1
        self.run_chain()
        torch.cuda.synchronize()
        steps, t0 = 0, time.perf_counter()
        while True:
            self.run_chain()
            steps += 1
            if steps >= min_steps and steps % 5 == 0:
                torch.cuda.synchronize()
                if time.perf_counter() - t0 >= min_seconds:
                    break
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el / (steps * reps * self.layers) * 1e6, steps * reps, el

    def eager_us_per_call(self, calls=2000):
        """Host cost of the product path: wall time per eager `layer(x)` call (Python -> ctypes -> one C call -> launch),
        back to back on the side stream with the GPU kept busy, so it is max(host time per call, device time per launch)."""
        with torch.cuda.stream(self.stream):
            for i in range(64):
                self.call(self.mods[i % self.layers])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(calls):
                self.call(self.mods[i % self.layers])
            t_host = time.perf_counter() - t0  # all calls ISSUED
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
        return t_host / calls * 1e6, t_all / calls * 1e6

    def kernel_us(self, samples):
        """Mean device duration of ONE launch from HIP events attached to individual launches (eager, same rotation)."""
        from gemlite_amd.bench_utils import HipEvents
        ev = HipEvents()
        pairs = [(ev.create(), ev.create()) for _ in range(samples)]
        with torch.cuda.stream(self.stream):
            # The sampled launches must run BACK TO BACK like the timed region (and like the run rocprofv3 sees): an eager
            # launch costs ~13 us of host time, so without a head start the GPU idles between 5-us kernels and each one pays a
            # wake-up inside its event pair (seen on one box: 6.25 us per event vs 4.78 us per launch chained).  A spin kernel
            # holds the stream while the host queues all the launches behind it.
            try:
                torch.cuda._sleep(int(samples * 25e-6 * 2.0e9))
            except Exception:
                pass
            for i, (a, b) in enumerate(pairs):
                self.lib.gemlite_hip_set_profile_events(a, b)
                self.call(self.mods[i % self.layers])
        torch.cuda.synchronize()
        durs = np.array([ev.elapsed_ms(a, b) * 1e3 for a, b in pairs])
        for a, b in pairs:
            ev.destroy(a)
            ev.destroy(b)
        durs = durs[np.isfinite(durs) & (durs > 0)]
        return float(durs.mean()) if durs.size else float("nan")

    def kernel_name(self):
        from gemlite_amd.core import _static_args
        lin = self.mods[0]
        a = _static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
        x = self.x if self.e2e else (self.x[0] if self.dt in PREQUANT else self.x)
        a.matmul_type = -1
        a.x = a.out = 0x1000
        a.M = x.shape[0]
        from gemlite_amd.dtypes import TORCH_TO_DTYPE
        a.input_dtype = lin.input_dtype.value if self.dt in MX else TORCH_TO_DTYPE[x.dtype].value
        a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = x.stride(0), x.stride(1), a.N, 1
        if self.e2e and x.shape[0] > 1:
            import gemlite_amd.core as core
            # one launch where the library quantises the rows inside the matmul launch (csrc/gl_coopquant.h) ...
            if core.FUSE_ACT_QUANT_ROWS and core.TUNING_OVERRIDE is None and not core.lookup_tuning(-1, a.M, a) and \
                    self.lib.gemlite_hip_query(ctypes.byref(a)) == 0:
                return self.lib.gemlite_hip_kernel_name(ctypes.byref(a)).decode()
            # ... else quantiser + matmul: name the matmul the quantised x reaches
            a.input_dtype = lin.input_dtype.value
            a.scales_x = 0x1000
        elif self.dt in PREQUANT and not self.e2e:
            a.scales_x = 0x1000
            if self.dt in MX:
                a.stride_sx_m = self.x[1].stride(0)
        import gemlite_amd.core as core
        t = core.TUNING_OVERRIDE or core.lookup_tuning(-1, a.M, a)
        if t:
            for i in range(4):
                a.tuning[i] = int(t[i])
        return self.lib.gemlite_hip_kernel_name(ctypes.byref(a)).decode()

    def roofline(self, samples=0, min_seconds=0.25):
        """One clock for every block: graph-replayed wall time per launch (>= 32 launches per replay).  `samples` > 0 adds the
        per-launch HIP-event figure as `event_us` (secondary)."""
        c_us, launches, el = self.chained_us_per_launch(min_seconds=min_seconds)
        out = {"workload": self.name, "bound": self.bound, "kernel": self.kernel_name(), "kernel_us": round(c_us, 3),
               "clock": f"hipGraph replay, {launches} launches in {el:.2f} s"}
        if samples > 0:
            try:
                e_us = self.kernel_us(samples)
                out["event_us"] = None if e_us != e_us else round(e_us, 3)
            except Exception:
                out["event_us"] = None
        if self.bound == "hbm":
            peak, unit, work = HBM_PEAK_GBS, "GB/s", self.bytes / 1e9
        else:
            peak = {"int8": INT8_MFMA_PEAK_TOPS, "fp8": INT8_MFMA_PEAK_TOPS, "fp8w8": INT8_MFMA_PEAK_TOPS, "mxa8": MXFP8_MFMA_PEAK_TFLOPS,
                    "mxa4": MXFP4_MFMA_PEAK_TFLOPS, "nva4": MFMA_PEAK_TFLOPS}.get(self.dt, MFMA_PEAK_TFLOPS)
            unit, work = "TFLOP/s", self.flops / 1e12
        ach = work / (c_us * 1e-6)
        out.update({"achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                    "algorithmic_bytes_per_launch": self.bytes, "flops_per_launch": self.flops})
        return out


def _committed(fname, name, kernel):
    """Per-workload figure from a committed profile summary under profiles/ (PMC passes are separate runs), or None.  Every entry is
    stamped (`_stamp[name] = {kernel, commit}`, written by scripts/pmc_official.py) with the planner's kernel name it was collected on:
    when the planner now picks another kernel for the workload the committed counter is STALE and the line says null instead."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", fname)))
        st = d.get("_stamp", {}).get(name)
        if st is None or st.get("kernel") != kernel:
            return None
        return d.get(name)
    except Exception:
        return None


def _traffic(name, kernel):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), or None (absent, or taken on another kernel)."""
    return _committed("pmc_traffic.json", name, kernel)


def _counter_commit(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("_stamp", {}).get(name, {}).get("commit")
    except Exception:
        return None


def event_clock_floor_us(lib, stream, samples=64):
    """An EMPTY 256 x 256 kernel timed with the same per-launch events."""
    from gemlite_amd.bench_utils import HipEvents
    ev = HipEvents()
    pairs = [(ev.create(), ev.create()) for _ in range(samples)]
    for a, b in pairs:
        lib.gemlite_hip_set_profile_events(a, b)
        lib.gemlite_hip_launch_noop(256, 256, stream.cuda_stream)
    torch.cuda.synchronize()
    d = np.array([ev.elapsed_ms(a, b) * 1e3 for a, b in pairs])
    for a, b in pairs:
        ev.destroy(a)
        ev.destroy(b)
    d = d[np.isfinite(d) & (d > 0)]
    return float(d.mean()) if d.size else float("nan")


def empty_launch_period_us(lib, stream, n=64, min_seconds=0.05):
    """Time per launch of an EMPTY 256 x 256 kernel, back to back inside a replayed hipGraph (the clock of every block of the line):
    the dependent-launch boundary alone."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(n):
            lib.gemlite_hip_launch_noop(256, 256, torch.cuda.current_stream().cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    reps, t0 = 0, time.perf_counter()
    while True:
        g.replay()
        reps += 1
        if reps % 10 == 0:
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= min_seconds:
                break
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * n) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="a16w4_4096_m1", choices=sorted(WORKLOADS))
    ap.add_argument("--single", action="store_true", help="only the named workload (no M=256 / trend / sustained blocks)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="profiler runs: short blocks (0.03 s each), sustained leg 0.2 s")
    ap.add_argument("--kernel-samples", type=int, default=256, help="launches timed individually for roofline")
    ap.add_argument("--full-out", default="", help="also write the verbose per-block records (clock strings, event timings, byte counts) to this file")
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

    name = args.workload
    N, K, nbits, group, M, dt, _, bound = WORKLOADS[name]
    main_run = Runner(name, device, lib, matmul_type=args.matmul_type, use_graph=not args.no_graph)
    layers = main_run.layers

    elapsed = timed_steps(rgroup, main_run.run_step, args.steps, args.warmup, device_sync=torch.cuda.synchronize, device=device)
    launches = args.steps * layers
    ms_per_step = elapsed / args.steps * 1e3
    gap_us = elapsed / launches * 1e6

    work = main_run.bytes / 1e9 if bound == "hbm" else main_run.flops / 1e12
    unit = "GB/s" if bound == "hbm" else "TFLOP/s"
    peak = HBM_PEAK_GBS if bound == "hbm" else {"int8": INT8_MFMA_PEAK_TOPS, "fp8": INT8_MFMA_PEAK_TOPS, "fp8w8": INT8_MFMA_PEAK_TOPS, "mxa8": MXFP8_MFMA_PEAK_TFLOPS,
                                                  "mxa4": MXFP4_MFMA_PEAK_TFLOPS}.get(dt, MFMA_PEAK_TFLOPS)
    value = whole_job_rate(layers * work, args.steps, world, elapsed)
    # the headline roofline IS the timed region: algorithmic bytes (flops) per launch / (elapsed / launches)
    ach = work / (gap_us * 1e-6)
    roof = {"bound": bound, "kernel": main_run.kernel_name(), "kernel_us": round(gap_us, 3),
            "clock": f"the timed region: {launches} launches in {elapsed * 1e3:.3f} ms of hipGraph replays" if main_run.graph is not None else "the timed region (eager launches)",
            "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "algorithmic_bytes_per_launch": main_run.bytes, "flops_per_launch": main_run.flops}
    if args.kernel_samples > 0:
        try:  # secondary: per-launch HIP events (their own floor printed next to them)
            e_us = main_run.kernel_us(min(args.kernel_samples, 1024))
            roof["event_us"] = None if e_us != e_us else round(e_us, 3)
            roof["event_clock_floor_us"] = round(event_clock_floor_us(lib, main_run.stream), 3)
        except Exception as e:
            print(f"[bench] per-kernel event timing unavailable: {e}", file=sys.stderr)
    try:  # the dependent-launch period of an EMPTY kernel inside a replayed graph, same process: what no kernel can go below
        e0 = empty_launch_period_us(lib, main_run.stream)
        roof["empty_launch_us"] = round(e0, 3)
        roof["kernel_us_minus_empty_launch"] = round(gap_us - e0, 3)
    except Exception as e:
        print(f"[bench] empty-launch period unavailable: {e}", file=sys.stderr)
    if bound == "hbm":
        roof["frac_vs_measured_copy_6290"] = round(roof["achieved"] / 6290.0, 4)
    else:
        roof["mfma_util"] = _committed("mfma_util.json", name, roof["kernel"])
    roof["traffic"] = _traffic(name, roof["kernel"])
    roof["traffic_commit"] = _counter_commit(name) if roof["traffic"] is not None else None

    metric = ("HBM GB/s (algorithmic bytes) vs roofline at M=1 [value], TFLOP/s vs bf16 MFMA roofline at M=256 [roofline.m256]; "
              "A16W4 gs=128 4096x4096" if name == "a16w4_4096_m1" else f"{unit} {name}")
    line = {
        "metric": metric, "value": round(value, 3), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dt, "data": "synthetic (seeded random W_q/scales/zeros/x, random-init)",
        "config": {"workload": f"A{4 if dt == 'mxa4' else (8 if dt in PREQUANT else 16)}W{nbits} gs={group} {N}x{K} M={M} {dt}; step = {layers} distinct layers "
                               f"(cache-cold rotation), {'hipGraph replay' if main_run.graph is not None else 'eager'}",
                   "layers_per_step": layers, "launches_per_step": layers, "replicas": world,
                   **({"tuning": args.tuning} if args.tuning else {}),
                   **({"matmul_type": args.matmul_type} if args.matmul_type else {})},
        "roofline": roof,
    }

    extras = name == "a16w4_4096_m1" and not args.single and not args.tuning and not args.matmul_type
    full = {}  # verbose blocks (--full-out); the printed line carries their compact form inside `roofline`
    if extras:
        def compact(b):
            c = {"kernel": b.get("kernel"), "kernel_us": b.get("kernel_us"), "frac": b.get("frac"), "traffic": b.get("traffic")}
            if b.get("bound") == "mfma":
                mu = b.get("mfma_util")
                c["mfma_util"] = mu.get("mfma_busy") if isinstance(mu, dict) else mu
            return c

        def block(wname, nl=None, samples=0, e2e=False):
            r = Runner(wname, device, lib, layers=nl, use_graph=not args.no_graph, e2e=e2e)
            out = r.roofline(0 if args.quick else samples, min_seconds=0.03 if args.quick else 0.25)
            if e2e:
                out["what"] = ("layer(x) on fp16 x, dynamic per-token quantisation included: " +
                               ("ONE fused launch" if r.M == 1 else "quantiser launch + matmul launch; kernel_us is the time of the pair"))
            out["traffic"] = _traffic(wname, out["kernel"])
            if out["bound"] == "mfma":
                out["mfma_util"] = _committed("mfma_util.json", wname, out["kernel"])
            del r
            torch.cuda.empty_cache()
            return out

        def add_group(key, blocks):
            """blocks: {label: verbose block}.  The compact form goes INSIDE `roofline` (the key the driver keeps whole)."""
            full[key] = blocks
            roof[key] = {k: compact(v) for k, v in blocks.items()}

        # every group is independent: one failing block must not take the M = 256 half of the metric with it
        def guarded(key, fn):
            try:
                add_group(key, fn())
            except Exception as e:
                roof[key] = {"error": f"{type(e).__name__}: {e}"[:160]}
                print(f"[bench] block {key} failed: {type(e).__name__}: {e}", file=sys.stderr)

        # Order of the groups = reverse order of importance: the driver keeps the TAIL of the printed line, so the bulky side groups go
        # first and the two halves of BASELINE's metric (the M = 1 core fields and `m256`) are re-appended at the very end of `roofline`.
        # block-scaled formats at decode-batch sizes (few-row scaled-MFMA kernel; NVFP4 on the fp16 tile kernel) and at M = 256 (unsplit
        # 64 x 64 scaled-MFMA tiles): matmul on pre-quantised x and layer(x) with the activation quantiser
        guarded("mx_fewrows", lambda: {**{w: block(w) for w in ("mx_a8w8_4096_m16", "mx_a4w4_4096_m16", "nvfp4_4096_m16", "nvfp4_4096_m256")},
                                       **{w + "_layer_e2e": block(w, e2e=True) for w in ("mx_a4w4_4096_m16", "nvfp4_4096_m16")}})
        guarded("mx_m256", lambda: {**{w: block(w) for w in ("mx_a8w8_4096_m256", "mx_a4w4_4096_m256")},
                                    **{w + "_layer_e2e": block(w, e2e=True) for w in ("mx_a8w8_4096_m256", "mx_a4w4_4096_m256")}})
        # BASELINE configs[3]: A8W8 int8 channel-wise 4096^2, M in {1, 16, 256}: the matmul alone on a pre-quantised x, and
        # (`*_layer_e2e`) layer(x) as the product runs it, dynamic activation quantisation included
        guarded("cfg4", lambda: {**{f"a8w8_int8_4096_m{m}": block(f"a8w8_4096_m{m}") for m in (1, 16, 32, 64, 256)},
                                 **{f"a8w8_int8_4096_m{m}_layer_e2e": block(f"a8w8_4096_m{m}", e2e=True) for m in (1, 16, 32, 64, 256)}})
        # BASELINE configs[4]: A16W2 g128 and FP8 x FP8, 16384^2, M in {1, 256}
        guarded("cfg5", lambda: {"a16w2_16384_m1": block("a16w2_16384_m1"), "a16w2_16384_m256": block("a16w2_16384_m256"),
                                 "fp8_16384_m1": block("fp8_16384_m1"), "fp8_16384_m256": block("fp8_16384_m256"),
                                 "fp8_16384_m1_layer_e2e": block("fp8_16384_m1", e2e=True),
                                 "fp8_16384_m256_layer_e2e": block("fp8_16384_m256", e2e=True)})
        # the large-M end of the same kernel family (north star: "tiled GEMM for large-M prefill") and the GEMV family on larger layers
        guarded("prefill_m2048", lambda: {"a16w4_8192_m2048": block("a16w4_8192_m2048", 4)})
        guarded("trend_m1", lambda: {"8192": block("a16w4_8192_m1"), "16384": block("a16w4_16384_m1")})
        # SURVEY §8(d) config 2 is "fp16 + bf16": the bf16 twin of the headline
        guarded("m1_bf16", lambda: {"a16w4_4096_m1_bf16": block("a16w4_4096_m1_bf16")})
        # decode-batch sizes of the headline layer (row a9: 2 <= M <= 64), layer(x) in fp16
        guarded("fewrows", lambda: {f"a16w4_4096_m{m}": block(f"a16w4_4096_m{m}") for m in (16, 32, 64)})
        # the M=256 half of the headline metric, same process, bf16
        guarded("m256", lambda: {"cfgA_4096": block("a16w4_4096_m256", samples=64), "cfgB_8192": block("a16w4_8192_m256", samples=32)})
        try:
            # the rotation size does not carry the headline: the same step over 32 layers (286 MB, rounds 1-3) and 64 (572 MB)
            r32 = block("a16w4_4096_m1", 32)
            roof["rotation_ab"] = {"layers32_286MB_us": r32["kernel_us"], "layers64_572MB_us": roof["kernel_us"]}
            h_us, e_us = main_run.eager_us_per_call()
            roof["eager"] = {"host_us_per_call": round(h_us, 3), "us_per_call_incl_device": round(e_us, 3)}
            full["eager"] = dict(roof["eager"], calls=2000, what="layer(x) eager, no graph: Python + ctypes + gemlite_hip_forward per call")
            # >= 6 s of back-to-back replays of the headline step
            us, nl, el = main_run.chained_us_per_launch(min_seconds=0.2 if args.quick else 6.0, min_steps=50)
            roof["sustained"] = {"seconds": round(el, 3), "value": round(main_run.bytes / 1e9 / (us * 1e-6), 3), "us_per_launch": round(us, 3)}
            full["sustained"] = dict(roof["sustained"], launches=nl, unit="GB/s")
        except Exception as e:
            print(f"[bench] eager / sustained legs failed: {type(e).__name__}: {e}", file=sys.stderr)
        # Key order of `roofline` (schema 6): the driver's parsed record keeps the FIRST 12 scalar keys of the object and the tail of the
        # printed line.  So the 12 scalars that carry both halves of BASELINE's metric come first — the M = 1 core (bound, achieved, peak,
        # unit, frac, traffic, kernel_us, empty_launch_us) and the M = 256 pair as flat scalars — then the other scalars, then the nested
        # groups in reverse order of importance (`fewrows` and `m256` last: they survive a tail cut).
        m256 = roof.get("m256", {})
        flat = {"m256_cfgA_us": m256.get("cfgA_4096", {}).get("kernel_us"), "m256_cfgA_frac": m256.get("cfgA_4096", {}).get("frac"),
                "m256_cfgB_us": m256.get("cfgB_8192", {}).get("kernel_us"), "m256_cfgB_frac": m256.get("cfgB_8192", {}).get("frac")}
        first = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_us", "empty_launch_us")
        ordered = {k: roof.get(k) for k in first}
        ordered.update(flat)
        ordered["schema"] = 6
        for k, v in roof.items():  # remaining scalars, then the nested groups
            if k not in ordered and not isinstance(v, dict):
                ordered[k] = v
        for k, v in roof.items():
            if k not in ordered and k not in ("fewrows", "m256"):
                ordered[k] = v
        for k in ("fewrows", "m256"):
            if k in roof:
                ordered[k] = roof[k]
        roof = ordered
        line["roofline"] = roof

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.torch_cpu_path import time_cpu_baseline
        r = time_cpu_baseline(M, N, K, nbits, group, budget_s=12.0)
        sec = r["sec_per_call"]
        cpu_val = (main_run.bytes / sec / 1e9) if bound == "hbm" else (main_run.flops / sec / 1e12)
        mm = r["matmul_only_sec"]
        line["cpu_baseline"] = {
            "value": round(cpu_val, 5), "unit": unit, "cores": r["threads"], "kind": "port",
            "sample": f"{r['calls']} calls unpack+dequant+matmul torch-CPU fp32, one {N}x{K} layer M={M}, {sec * 1e3:.1f} ms/call"}
        full["cpu_baseline"] = dict(line["cpu_baseline"], host_cpus=r["host_cpus"],
                                    thread_sweep_ms={str(k): round(v * 1e3, 2) for k, v in r["sweep"].items()},
                                    matmul_only={"ms_per_call": round(mm * 1e3, 3), "threads": r["matmul_only_threads"],
                                                 "note": "x @ W.T on the pre-dequantised fp32 W (64 MiB read per call)"})
    line["roofline"] = line.pop("roofline")  # last key of the line: its tail (fewrows, m256, the M = 1 core) is what a truncated record keeps
    if rank == 0:
        if args.full_out:
            with open(args.full_out, "w") as f:
                json.dump({"line": line, "blocks": full}, f, indent=1)
        print(json.dumps(line), flush=True)
    rgroup.close()


if __name__ == "__main__":
    main()
