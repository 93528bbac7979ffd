#!/usr/bin/env python3
"""Generate tests/golden/mx.npz: what THE REFERENCE produces for the block-scaled (MXFP8 / MXFP4 / NVFP4) formats.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  Same process-local device fakes as
oracle/gen_golden.py.  What runs here, on CPU:
  * WeightQuantizerMXFP.quantize_mxfp8 / _mxfp4 / _nvfp4 (torch code; `torch.compile` is disabled, and the per-CUDA-device
    lookup tables of quant_utils.py:27-68 — empty lists without a GPU — are replaced by CPU tensors);
  * the processors A16W8/W4_MXFP, A8W8/W4_MXFP_dynamic, A4W4_MXFP_dynamic, A4W4_NVFP_dynamic (host code: pack());
  * the activation quantisers scale_activations_mxfp8 / mxfp4 / nvfp4 (= the *_triton_v2 kernels) under TRITON_INTERPRET=1.
    The interpreter's own fp32 -> fp8 conversion is not round-to-nearest-even (it adds the cut-off bit without carrying
    into the exponent: 126.9 -> 64); it is replaced — in the interpreter, not in the reference — by torch's conversion,
    which is what the compiled kernel's `.to(tl.float8e4nv)` does on hardware;
  * the MX matmul kernels (tl.dot_scaled) are tried under the interpreter and recorded when they run.

Usage:  python oracle/gen_golden_mx.py [--out tests/golden]
"""
import argparse
import os
import sys

os.environ["TORCHDYNAMO_DISABLE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # sets TRITON_INTERPRET and provides the device fakes

import numpy as np
import torch


class _AnyIndex:
    """stands in for the reference's per-device list of lookup tensors: every index (None on CPU) -> the CPU tensor"""

    def __init__(self, t):
        self.t = t

    def __getitem__(self, _):
        return self.t


def _patch_interpreter_fp8():
    import triton.language as tl
    import triton.runtime.interpreter as interp

    orig = interp._convert_float

    def cf(inp, in_dt, out_dt, rounding):
        if in_dt == tl.float32 and out_dt == tl.float8e4nv:
            t = torch.from_numpy(np.ascontiguousarray(inp).astype(np.float32)).to(torch.float8_e4m3fn)
            return t.view(torch.uint8).numpy().reshape(inp.shape)
        return orig(inp, in_dt, out_dt, rounding)

    interp._convert_float = cf

    # `a_dtype: tl.constexpr = "e4m3"` inside a jitted function is a plain str under the interpreter; tl.dot_scaled wants
    # constexpr operands for its format arguments
    orig_ds = tl.dot_scaled

    def dot_scaled(lhs, lhs_scale, lhs_format, rhs, rhs_scale, rhs_format, *a, **k):
        if isinstance(lhs_format, str):
            lhs_format = tl.constexpr(lhs_format)
        if isinstance(rhs_format, str):
            rhs_format = tl.constexpr(rhs_format)
        return orig_ds(lhs, lhs_scale, lhs_format, rhs, rhs_scale, rhs_format, *a, **k)

    tl.dot_scaled = dot_scaled


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    G._fake_device_and_import()
    _patch_interpreter_fp8()
    from gemlite import helper as H
    from gemlite import quant_utils as Q

    f32 = torch.float32
    Q.fp4_values = _AnyIndex(torch.tensor([0, 0.5, 1, 1.5, 2, 3, 4, 6, -0.0, -0.5, -1, -1.5, -2, -3, -4, -6], dtype=f32))
    Q.fp4_p_vals = _AnyIndex(torch.tensor([0, 0.5, 1, 1.5, 2, 3, 4, 6], dtype=f32))
    Q.fp4_thresholds = _AnyIndex(torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0], dtype=f32))
    Q.thr_pos = _AnyIndex(torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 7.0], dtype=f32))
    fp8 = torch.float8_e4m3fn
    g = torch.Generator().manual_seed(2468)
    blob, notes = {}, []

    # ------------------------------------------------------------------ 1. weight quantiser
    N, K = 48, 256
    W = (torch.randn(N, K, generator=g) / 10).to(torch.bfloat16)
    W[3, 32:64] = 0          # an all-zero block
    W[5, 0] = 3.0            # an outlier block
    W[7, 64:96] = 0.375 / 8  # values that sit exactly on rounding midpoints after scaling
    blob["wq_in_W"] = G._np(W)
    wq = Q.WeightQuantizerMXFP(compute_dtype=torch.bfloat16, device="cpu")
    for name, fn in (("mxfp8", lambda: wq.quantize_mxfp8(W, index=True, mx_fp8_dtype=fp8)),
                     ("mxfp4", lambda: wq.quantize_mxfp4(W, index=True)),
                     ("mxfp4_w1", lambda: wq.quantize_mxfp4(W, window_size=1, index=True)),
                     ("nvfp4", lambda: wq.quantize_nvfp4(W, index=True)),
                     ("nvfp4_w1", lambda: wq.quantize_nvfp4(W, window_size=1, index=True))):
        q, s = fn()
        blob[f"wq_{name}_q"] = q.contiguous().view(torch.uint8).numpy()
        blob[f"wq_{name}_s"] = s.contiguous().view(torch.uint8).numpy()
    print("weight quantiser: 5 cases")

    # ------------------------------------------------------------------ 2. processors (host side: pack())
    lin = torch.nn.Linear(K, N, bias=True, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(W)
        lin.bias.copy_((torch.randn(N, generator=g) / 10).to(torch.bfloat16))
    blob["proc_in_bias"] = G._np(lin.bias.data)
    procs = {
        "a16w8_mxfp": lambda: H.A16W8_MXFP(device="cpu", dtype=torch.bfloat16),
        "a16w4_mxfp": lambda: H.A16W4_MXFP(device="cpu", dtype=torch.float16),
        "a8w8_mxfp_dyn_post": lambda: H.A8W8_MXFP_dynamic(device="cpu", dtype=torch.bfloat16, post_scale=True, fp8=fp8),
        "a8w8_mxfp_dyn_micro": lambda: H.A8W8_MXFP_dynamic(device="cpu", dtype=torch.bfloat16, post_scale=False, fp8=fp8),
        "a8w4_mxfp_dyn": lambda: H.A8W4_MXFP_dynamic(device="cpu", dtype=torch.bfloat16, post_scale=False, fp8=fp8),
        "a4w4_mxfp_dyn": lambda: H.A4W4_MXFP_dynamic(device="cpu", dtype=torch.bfloat16),
        "a4w4_nvfp_dyn": lambda: H.A4W4_NVFP_dynamic(device="cpu", dtype=torch.float16),
    }
    names = []
    for name, make in procs.items():
        p = make()
        if hasattr(p, "mx_fp8_dtype"):
            p.mx_fp8_dtype = fp8  # the reference's HIP default is MI300X's e4m3fnuz
        layer = p.from_linear(lin, del_orig=False)
        wq_t, sc_t = layer.W_q.data, layer.scales.data
        blob[f"proc_{name}_W_q"] = wq_t.contiguous().view(torch.uint8).numpy()
        blob[f"proc_{name}_W_q_shape_stride"] = np.array(list(wq_t.shape) + list(wq_t.stride()), dtype=np.int64)
        blob[f"proc_{name}_scales"] = sc_t.contiguous().view(torch.uint8).numpy()
        blob[f"proc_{name}_scales_shape_stride"] = np.array(list(sc_t.shape) + list(sc_t.stride()), dtype=np.int64)
        blob[f"proc_{name}_meta"] = np.array(layer.get_meta_args(), dtype=np.int64)
        blob[f"proc_{name}_bias"] = G._np(layer.bias.data)
        names.append(name)
    blob["proc_names"] = np.array(names)
    print("processors:", len(names))

    # ------------------------------------------------------------------ 3. activation quantisers (Triton interpreter)
    for tag, tdt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        M, Kx = 5, 256
        x = (torch.randn(M, Kx, generator=g) * 0.7).to(tdt)
        x[1, 0:32] = 0
        x[2, 5] = 40.0
        x[3, 64:96] = (torch.arange(32) - 16).to(tdt) * 0.046875   # many exact midpoints (0.75 / 16 steps)
        x[4, 96:112] = -x[4, 96:112].abs() * 1e-3                    # tiny negatives: the "-0" code
        blob[f"act_{tag}_x"] = G._np(x)
        for name, fn in (("mxfp8", lambda t: Q.scale_activations_mxfp8(t, w_dtype=fp8)),
                         ("mxfp4", Q.scale_activations_mxfp4), ("nvfp4", Q.scale_activations_nvfp4)):
            y, s = fn(x.clone())
            blob[f"act_{tag}_{name}_y"] = y.contiguous().view(torch.uint8).numpy()
            blob[f"act_{tag}_{name}_s"] = s.contiguous().view(torch.uint8).numpy()
            if name == "mxfp8":  # the interpreter leaves rows M..M_pad of `scales` untouched only where no program ran
                notes.append(f"{tag} {name}: y {tuple(y.shape)} scales {tuple(s.shape)}")
    print("activation quantisers: 6 cases")

    # ------------------------------------------------------------------ 4. MX matmul through the reference kernels
    ran = []
    try:
        from gemlite import DType  # noqa: F401

        xs = {}
        for M in (1, 4, 16):
            xs[M] = (torch.randn(M, K, generator=g) / 4).to(torch.bfloat16)
            blob[f"mm_x_{M}"] = G._np(xs[M])
        for name in ("a16w8_mxfp", "a8w8_mxfp_dyn_micro", "a8w4_mxfp_dyn", "a4w4_mxfp_dyn"):
            p = procs[name]()
            if hasattr(p, "mx_fp8_dtype"):
                p.mx_fp8_dtype = fp8
            layer = p.from_linear(lin, del_orig=False)
            for M in (1, 4, 16):
                try:
                    xin = xs[M] if name != "a16w4_mxfp" else xs[M].to(torch.float16)
                    y = layer.forward_manual(xin, matmul_type="GEMM_SPLITK")
                    blob[f"mm_{name}_{M}"] = G._np(y.float())
                    ran.append(f"{name}:{M}")
                except Exception as e:  # noqa: BLE001
                    notes.append(f"matmul {name} M={M} did not run under the interpreter: {type(e).__name__}: {str(e)[:300]}")
                    break
    except Exception as e:  # noqa: BLE001
        notes.append(f"matmul section failed: {type(e).__name__}: {str(e)[:200]}")
    blob["mm_ran"] = np.array(ran if ran else ["none"])
    blob["notes"] = np.array(notes if notes else ["-"])
    print("matmul cases that ran under the interpreter:", ran)
    for n in notes:
        print("note:", n)

    out = os.path.join(os.path.abspath(args.out), "mx.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
