#!/usr/bin/env python3
"""Generate tests/golden/*.npz from THE REFERENCE ITSELF (build container only).

TEST INFRASTRUCTURE.  Run once in the build container (needs /root/reference, which does not
exist on the GPU box); the resulting fixtures are committed and are what pins the oracle
(oracle/gemlite_oracle.py) and the HIP path to the reference's behaviour.

How the reference is executed without a GPU:
  * TRITON_INTERPRET=1 runs its Triton kernels on CPU tensors;
  * `import gemlite` probes the GPU at import (triton_kernels/utils.py:130,182; core.py:640), so
    three PROCESS-LOCAL monkeypatches of torch/triton (never of the reference) fake a device;
  * the interpreter cannot convert a Python bool kernel argument (`load_scales_as_block`,
    gemm_splitK_kernels.py) — patched in triton's interpreter, again not in the reference.
Kernels that run this way: GEMV_REVSPLITK and GEMM_SPLITK (fp32 `tl.dot` accumulation, same
arithmetic as GEMM), and scale_activations_per_token_triton.  GEMM needs a constexpr-float
multiply the interpreter lacks; GEMV needs a CUDA device index; GEMV_SPLITK's default config
is numerically wrong for grouped weights in the reference.

Usage:  python oracle/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"
import numpy as np
import torch

import triton
import triton.language as tl

REF = "/root/reference"


@triton.jit
def _floor(x):  # stand-in for libdevice.floor under the interpreter (see _fake_device_and_import)
    return tl.floor(x)


def _fake_device_and_import():
    class _Props:
        name = "AMD Instinct MI355X"
        multi_processor_count = 256
        total_memory = 288 << 30

    torch.cuda.get_device_properties = lambda *a, **k: _Props()
    torch.cuda.device_count = lambda: 0
    import triton

    class _Target:
        backend = "hip"
        arch = "gfx950"
        warp_size = 64

    class _Utils:
        def get_device_properties(self, *a):
            return {"max_shared_mem": 163840, "multiprocessor_count": 256}

    class _Driver:
        utils = _Utils()

        def get_current_target(self):
            return _Target()

        def get_current_device(self):
            return 0

        def get_device_interface(self):
            return torch.cuda

        def get_active_torch_device(self):
            return torch.device("cpu")

        def get_benchmarker(self):
            return lambda *a, **k: 0.0

        def get_empty_cache_for_benchmark(self):
            return torch.empty(1)

        def clear_cache(self, c):
            pass

    triton.runtime.driver.set_active(_Driver())
    import triton.language as tl
    import triton.runtime.interpreter as interp

    _orig = interp._implicit_cvt

    def _cvt(arg):
        if isinstance(arg, bool):
            return tl.core.tensor(interp.TensorHandle(np.array([arg], dtype=np.bool_), tl.int1), tl.int1)
        return _orig(arg)

    interp._implicit_cvt = _cvt

    # libdevice externs return None under the interpreter; the reference's AMD rounding is
    # libdevice.floor(x + 0.5) (quant_utils.py:259-266) -> route it to tl.floor.
    from triton.language.extra import libdevice

    libdevice.floor = _floor
    sys.path.insert(0, REF)
    import gemlite

    gemlite.reset_config()
    gemlite.set_autotune(False)
    return gemlite


def _np(t):
    """tensor -> ndarray preserving bits (bf16/fp8 stored as raw integer views)."""
    if t is None:
        return None
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy()
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return t.view(torch.uint8).numpy()
    return t.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    gemlite = _fake_device_and_import()
    from gemlite import DType, GemLiteLinear
    from gemlite.bitpack import pack_weights_over_cols_torch
    from gemlite.dtypes import TORCH_TO_DTYPE

    g = torch.Generator().manual_seed(1234)

    def rand_wq(N, K, nb):
        return torch.randint(0, 2 ** nb, (N, K), generator=g, dtype=torch.int32).to(torch.uint8)

    # ---------------------------------------------------------------- 1. raw bit packing
    pk = {}
    for nb in (1, 2, 4, 8):
        for pb in (8, 16, 32):
            if pb < nb:
                continue
            W = rand_wq(24, 192, nb)
            packed, e = pack_weights_over_cols_torch(W, nb, pb, True)
            pk[f"in_{nb}_{pb}"] = _np(W)
            pk[f"out_{nb}_{pb}"] = _np(packed.contiguous())
            pk[f"e_{nb}_{pb}"] = np.int64(e)
    np.savez_compressed(os.path.join(out_dir, "bitpack.npz"), **pk)
    print("bitpack:", len(pk) // 3, "cases")

    # ---------------------------------------------------------------- 2. pack() + forward()
    cases = []

    def add_case(name, *, nb, gs, N, K, in_dt, out_dt, tdt, scales_kind, zeros_kind, fma=True, scaled_act=False,
                 pb=None, Ms=(1, 4), unpacked=None, xkind="randn", scales_f32=False):
        cases.append(dict(name=name, nb=nb, gs=gs, N=N, K=K, in_dt=in_dt, out_dt=out_dt, tdt=tdt,
                          scales_kind=scales_kind, zeros_kind=zeros_kind, fma=fma, scaled_act=scaled_act, pb=pb,
                          Ms=Ms, unpacked=unpacked, xkind=xkind, scales_f32=scales_f32))

    H, B = torch.float16, torch.bfloat16
    add_case("a16w4_g128_fma_fp16", nb=4, gs=128, N=64, K=512, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="tensor", Ms=(1, 4, 16))
    add_case("a16w4_g128_fma_bf16", nb=4, gs=128, N=64, K=512, in_dt=DType.BF16, out_dt=DType.BF16, tdt=B,
             scales_kind="group", zeros_kind="tensor", Ms=(1, 4))
    add_case("a16w4_g64_nofma_fp16", nb=4, gs=64, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="tensor", fma=False)
    add_case("a16w2_g64_fma_fp16", nb=2, gs=64, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="tensor")
    add_case("a16w1_g32_fma_fp16", nb=1, gs=32, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="tensor")
    add_case("a16w8_g128_fma_fp16", nb=8, gs=128, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="tensor")
    add_case("a16w4_g128_sym_fp16", nb=4, gs=128, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="none")
    add_case("a16w4_g128_intzero_fp16", nb=4, gs=128, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="int")
    add_case("a16w4_channel_intzero_fp16", nb=4, gs=256, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="channel", zeros_kind="int")
    add_case("a16w4_channel_tzero_fp16", nb=4, gs=256, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="channel", zeros_kind="tensor")
    add_case("a16w4_g128_pack8_fp16", nb=4, gs=128, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="group", zeros_kind="tensor", pb=8)
    add_case("a8w4_int8_channel_intzero", nb=4, gs=256, N=64, K=256, in_dt=DType.INT8, out_dt=DType.FP32, tdt=H,
             scales_kind="channel", zeros_kind="int", xkind="int8")
    add_case("a8w4_int8_dyn_intzero", nb=4, gs=128, N=64, K=256, in_dt=DType.INT8, out_dt=DType.FP32, tdt=H,
             scales_kind="none", zeros_kind="int", scaled_act=True)
    # fp32 channel scales like helper.py:459 (fp16 meta overflows: int32 acc -> fp16 = inf in the reference)
    add_case("a8w8_int8_dyn_channel", nb=8, gs=256, N=64, K=256, in_dt=DType.INT8, out_dt=DType.FP32, tdt=H,
             scales_kind="channel", zeros_kind="int", scaled_act=True, scales_f32=True)
    add_case("a16w16_fp16_unpacked", nb=16, gs=None, N=64, K=256, in_dt=DType.FP16, out_dt=DType.FP16, tdt=H,
             scales_kind="none", zeros_kind="none", unpacked="fp16")
    add_case("a8w8_int8_unpacked_dyn_channel", nb=8, gs=256, N=64, K=256, in_dt=DType.INT8, out_dt=DType.FP16, tdt=H,
             scales_kind="channel", zeros_kind="none", scaled_act=True, unpacked="int8", scales_f32=True)
    # (no fp8 *dynamic* forward golden: the interpreter's fp32->fp8 store drops the rounding carry into
    #  the exponent (126.9 -> 64 instead of 128), an artefact of the CPU interpreter, not of the reference)
    add_case("a8w8_fp8_unpacked_plain", nb=8, gs=None, N=64, K=256, in_dt=DType.FP8, out_dt=DType.FP16, tdt=H,
             scales_kind="none", zeros_kind="none", unpacked="fp8", xkind="fp8")

    blob = {}
    names = []
    for c in cases:
        N, K, nb, gs = c["N"], c["K"], c["nb"], c["gs"]
        tdt = c["tdt"]
        if c["unpacked"] == "fp16":
            W_in = (torch.randn(N, K, generator=g) / 10).to(tdt)
        elif c["unpacked"] == "fp8":
            W_in = (torch.randn(N, K, generator=g) * 50).to(torch.float8_e4m3fn)
        elif c["unpacked"] == "int8":
            W_in = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int32).to(torch.int8)
        else:
            W_in = rand_wq(N, K, nb)
        ng = 1 if gs is None else N * K // gs
        if c["scales_kind"] == "group":
            scales = (torch.rand(ng, 1, generator=g) * 0.01 + 0.001).to(tdt)
        elif c["scales_kind"] == "channel":
            scales = (torch.rand(N, 1, generator=g) * 0.01 + 0.001).to(torch.float32 if c["scales_f32"] else tdt)
        else:
            scales = None
        if c["zeros_kind"] == "tensor":
            nz = ng if c["scales_kind"] == "group" else N
            zeros = (torch.rand(nz, 1, generator=g) * (2 ** nb - 1)).to(tdt)
        elif c["zeros_kind"] == "int":
            zeros = (2 ** nb) // 2  # keeps q - z inside int8 for the int8 x 8-bit case
        else:
            zeros = None
        lin = GemLiteLinear(nb, gs, K, N, c["in_dt"], c["out_dt"], scaled_activations=c["scaled_act"])
        lin.pack(W_in, scales, zeros, None, fma_mode=c["fma"], packing_bitwidth=c["pb"])
        if c["name"] == "a8w4_int8_dyn_intzero":
            lin.meta_dtype = DType.FP32  # as the reference test does (test_gemlitelineartriton.py:194)
        nm = c["name"]
        names.append(nm)
        blob[nm + "/W_in"] = _np(W_in)
        blob[nm + "/scales_in"] = _np(scales) if scales is not None else np.zeros(0)
        blob[nm + "/zeros_in"] = _np(zeros) if isinstance(zeros, torch.Tensor) else np.array(
            [] if zeros is None else [zeros], dtype=np.int64)
        blob[nm + "/cfg"] = np.array([nb, -1 if gs is None else gs, N, K, c["in_dt"].value, c["out_dt"].value,
                                      int(c["fma"]), int(c["scaled_act"]), -1 if c["pb"] is None else c["pb"],
                                      TORCH_TO_DTYPE[tdt].value, {"none": 0, "int": 1, "tensor": 2}[c["zeros_kind"]],
                                      {"none": 0, "group": 1, "channel": 2}[c["scales_kind"]]], dtype=np.int64)
        blob[nm + "/W_q"] = _np(lin.W_q.data)
        blob[nm + "/W_q_stride"] = np.array(lin.W_q.stride(), dtype=np.int64)
        blob[nm + "/scales"] = _np(lin.scales.data)
        blob[nm + "/zeros"] = _np(lin.zeros.data)
        blob[nm + "/meta_args"] = np.array(lin.get_meta_args(), dtype=np.int64)
        print(nm, "meta", lin.get_meta_args(), "W_q", tuple(lin.W_q.shape), lin.W_q.dtype, tuple(lin.W_q.stride()))
        for M in c["Ms"]:
            if c["xkind"] == "int8":
                x = torch.randint(-10, 10, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
            elif c["xkind"] == "fp8":
                x = (torch.randn(M, K, generator=g) / 10).to(torch.float8_e4m3fn)
            else:
                x = (torch.randn(M, K, generator=g) / 10).to(tdt)
            blob[f"{nm}/x_M{M}"] = _np(x)
            for mt in (["GEMV_REVSPLITK"] if M == 1 else []) + ["GEMM_SPLITK"]:
                if mt == "GEMV_REVSPLITK" and (nb >= 8 or c["unpacked"]):
                    continue  # reference routes 8-bit to GEMV_SPLITK (core.py:105)
                if c["in_dt"] == DType.BF16:
                    continue  # interpreter artefact: bf16 kernels return non-finite garbage on CPU
                if mt == "GEMV_REVSPLITK" and gs is not None and gs < 64:
                    continue  # default (autotune off) config has 2*BLOCK_K = 64 > group_size: the
                    #           reference itself is wrong there (same class as SURVEY App. B.3)
                try:
                    y = lin.forward_manual(x, mt)
                    blob[f"{nm}/y_{mt}_M{M}"] = _np(y.float())
                except Exception as ex:  # record which reference kernels cannot run here
                    print("   ", mt, "M", M, "FAILED under interpreter:", type(ex).__name__, str(ex)[:120])
    blob["names"] = np.array(names)
    np.savez_compressed(os.path.join(out_dir, "pack_forward.npz"), **blob)
    print("pack_forward:", len(names), "cases")

    # ---------------------------------------------------------------- 3. activation quant
    from gemlite.quant_utils import scale_activations_per_token_torch, scale_activations_per_token_triton

    # the torch spec function is wrapped in torch.compile; call the undecorated python function
    torch_spec = getattr(scale_activations_per_token_torch, "_torchdynamo_orig_callable",
                         getattr(scale_activations_per_token_torch, "__wrapped__", scale_activations_per_token_torch))
    aq = {}
    for tag, dt in (("int8", torch.int8), ("fp8e4", torch.float8_e4m3fn), ("fp8e5", torch.float8_e5m2)):
        for M, K in ((1, 256), (5, 384)):
            x = (torch.randn(M, K, generator=g) / 7).to(torch.float16)
            x[0, 3] = 0.0
            if tag == "int8":  # the Triton kernel (AMD rounding floor(x+0.5)) — what runs on MI355X
                xq, s = scale_activations_per_token_triton(x, dt)
            else:  # fp8: the reference's torch spec (exact RNE cast); the interpreter's fp8 store is buggy
                xq, s = torch_spec(x.clone(), dt)
            aq[f"{tag}_x_{M}_{K}"] = _np(x)
            aq[f"{tag}_q_{M}_{K}"] = _np(xq.float())
            aq[f"{tag}_s_{M}_{K}"] = _np(s)
    np.savez_compressed(os.path.join(out_dir, "act_quant.npz"), **aq)
    print("act_quant:", len(aq) // 3, "cases")


if __name__ == "__main__":
    main()
