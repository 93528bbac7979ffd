"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A numpy (float64) restatement of the reference algorithm for the hot path
``GemLiteLinear.pack() -> forward()``.  Nothing in ``gemlite_amd/`` may import this file:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
only as the checker.  The product path has no CPU fallback.

Parity pinning: this oracle is checked (tests/test_oracle_golden.py) against
``tests/golden/*.npz`` which hold OUTPUTS OF THE REFERENCE ITSELF — its ``pack()`` host code
and its Triton kernels executed under ``TRITON_INTERPRET=1`` in the build container
(generator: oracle/gen_golden.py).  So parity is pinned, not "unpinned".

Each function cites the reference file:line it restates (paths relative to /root/reference).
torch is used here only to move bf16 / fp8 bit patterns into float32 (numpy has neither).
"""
from __future__ import annotations

import numpy as np

try:  # torch only for dtype conversion of bf16 / fp8 payloads
    import torch
except Exception:  # pragma: no cover
    torch = None

# dtype codes: gemlite/dtypes.py:8-29
FP32, FP16, BF16, FP8E4, INT8, UINT8, INT32 = 0, 1, 2, 3, 4, 5, 6
FP8E5 = 8

_PACK_NP = {8: np.uint8, 16: np.int16, 32: np.int32, 64: np.int64}


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def to_f64(t) -> np.ndarray:
    """torch tensor (any float/int dtype incl. bf16/fp8) or ndarray -> float64 ndarray."""
    if torch is not None and isinstance(t, torch.Tensor):
        t = t.detach().cpu()
        if t.dtype in (torch.bfloat16, torch.float8_e4m3fn, torch.float8_e5m2, torch.float16):
            t = t.to(torch.float32)
        return t.numpy().astype(np.float64)
    return np.asarray(t).astype(np.float64)


def round_to_dtype(a: np.ndarray, code: int) -> np.ndarray:
    """Round a float64 array to the value grid of dtype `code` (returns float64)."""
    if code == FP32:
        return a.astype(np.float32).astype(np.float64)
    if code == FP16:
        return a.astype(np.float16).astype(np.float64)
    if code in (BF16, FP8E4, FP8E5):
        tdt = {BF16: torch.bfloat16, FP8E4: torch.float8_e4m3fn, FP8E5: torch.float8_e5m2}[code]
        return torch.from_numpy(a.astype(np.float32)).to(tdt).to(torch.float32).numpy().astype(np.float64)
    if code == INT8:
        return np.clip(np.rint(a), -128, 127)
    if code == INT32:
        return np.rint(a)
    raise ValueError(f"unsupported dtype code {code}")


# --------------------------------------------------------------------------------------
# bit packing  (gemlite/bitpack.py:36-60 pack_weights_over_cols_torch; core.py:384-398,478-480)
# --------------------------------------------------------------------------------------
def pack_over_cols(W_q: np.ndarray, W_nbits: int, pack_bits: int = 32) -> np.ndarray:
    """W_q[N,K] (values < 2^b) -> packed[K/e, N] contiguous, e = pack_bits // W_nbits.

    word(n, j) = OR_i W_q[n, j*e + i] << (b*i)  (element i at bits [b*i, b*i+b), LSB first),
    then transposed so memory is N-contiguous.
    """
    assert pack_bits in _PACK_NP and W_nbits in (1, 2, 4, 8)
    e = pack_bits // W_nbits
    N, K = W_q.shape
    assert K % e == 0
    acc = np.zeros((N, K // e), dtype=np.uint64)
    w = W_q.astype(np.uint64).reshape(N, K // e, e)
    for i in range(e):
        acc |= w[:, :, i] << np.uint64(W_nbits * i)
    if pack_bits < 64:
        acc &= np.uint64((1 << pack_bits) - 1)
    words = acc.astype({8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}[pack_bits])
    words = words.view(_PACK_NP[pack_bits])  # reinterpret as the signed storage dtype
    return np.ascontiguousarray(words.T)


def unpack_over_cols(packed: np.ndarray, W_nbits: int, pack_bits: int = 32) -> np.ndarray:
    """packed[K/e, N] -> W_q[N, K] uint8.  element k of column n =
    (packed[k//e, n] >> ((k%e)*b)) & (2^b-1)   (gemm_kernels.py:327-328, utils.py:70-71)."""
    e = pack_bits // W_nbits
    Kp, N = packed.shape
    u = packed.astype({8: np.uint8, 16: np.int16, 32: np.int32, 64: np.int64}[pack_bits])
    u = u.view({8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}[pack_bits]).astype(np.uint64)
    out = np.empty((Kp, e, N), dtype=np.uint8)
    mask = np.uint64((1 << W_nbits) - 1)
    for i in range(e):
        out[:, i, :] = ((u >> np.uint64(W_nbits * i)) & mask).astype(np.uint8)
    return np.ascontiguousarray(out.reshape(Kp * e, N).T)


# --------------------------------------------------------------------------------------
# pack(): metadata layout + mode selection  (gemlite/core.py:336-519)
# --------------------------------------------------------------------------------------
def resolve_modes(*, has_scales: bool, scales_numel: int, zeros_kind: str, out_features: int,
                  scaled_activations: bool, fma_mode: bool = True):
    """Return (W_group_mode, channel_scale_mode, zeros_is_folded).

    zeros_kind: 'none' | 'tensor' | 'int'.  Restates core.py:408-464.
    """
    W_group_mode, channel_scale_mode = -1, 0
    if (not has_scales) and zeros_kind == "none":  # :412-416
        W_group_mode = 0
        channel_scale_mode = 2 if scaled_activations else 0
    channelwise = has_scales and scales_numel == out_features  # :425
    folded = False
    if zeros_kind == "none":  # :428-430
        W_group_mode = 2 if has_scales else 0
    elif zeros_kind == "tensor":  # :433-439
        if fma_mode and not channelwise:
            W_group_mode, folded = 4, True
        else:
            W_group_mode = 3
    else:  # integer zero :440-445
        W_group_mode = 3 if has_scales else 1
    if (not scaled_activations) and channelwise:  # :450-452
        channel_scale_mode = 1
        W_group_mode = 1 if zeros_kind != "none" else 0
    if scaled_activations and not channelwise:  # :455-456
        channel_scale_mode = 2
    if scaled_activations and channelwise:  # :459-461
        channel_scale_mode = 3
        W_group_mode = 1 if zeros_kind != "none" else 0
    return W_group_mode, channel_scale_mode, folded


def layout_meta(scales: np.ndarray | None, zeros, out_features: int, folded: bool, meta_code: int):
    """scales.view(N,-1).t() -> [K/g, N]; zeros likewise, or z' = round_meta(-z*s) when folded
    (core.py:419-420, 433-439)."""
    s_l = None if scales is None else np.ascontiguousarray(np.asarray(scales, np.float64).reshape(out_features, -1).T)
    z_l = None
    if isinstance(zeros, np.ndarray):
        z = np.asarray(zeros, np.float64)
        if folded:
            zf = (-(z.astype(np.float32)) * np.asarray(scales, np.float64).astype(np.float32)).astype(np.float64)
            z = round_to_dtype(zf, meta_code)
        z_l = np.ascontiguousarray(z.reshape(out_features, -1).T)
    elif zeros is not None:
        z_l = np.array([int(zeros)], dtype=np.float64)
    return s_l, z_l


# --------------------------------------------------------------------------------------
# dequantize + matmul  (gemlite/triton_kernels/utils.py:57-89; gemm_kernels.py:347-413;
#                       core.py:192-193 for bias)
# --------------------------------------------------------------------------------------
def dequantize(q_kn: np.ndarray, scales_gn, zeros_gn, group_size: int, W_group_mode: int,
               zero_is_scalar: bool = False, meta_code: int | None = None) -> np.ndarray:
    """q_kn: [K,N] unpacked integer (or float, for unpacked weights) values -> W[K,N] float64.

    meta_code=None: exact real arithmetic on the STORED scales/zeros (the parity target).
    meta_code=FP16/BF16: additionally round every intermediate like the reference kernels do
    (`b.to(meta)`, fp16 subtract/multiply, single-rounding fma) — used to calibrate tolerances.
    """
    K, N = q_kn.shape
    q = q_kn.astype(np.float64)
    rnd = (lambda a: a) if meta_code is None else (lambda a: round_to_dtype(a, meta_code))

    def expand(m):
        if m is None:
            return None
        m = np.asarray(m, np.float64)
        if m.size == 1:
            return m.reshape(1, 1)
        if m.ndim == 1 or m.shape[0] == 1:  # channel-wise [N] / [1,N]
            return m.reshape(1, N)
        return np.repeat(m, group_size, axis=0)[:K]

    s, z = expand(scales_gn), expand(zeros_gn)
    if W_group_mode == 0:
        return q
    if W_group_mode == 1:  # b.to(meta) - zeros
        return rnd(rnd(q) - z)
    if W_group_mode == 2:  # b.to(meta) * scales
        return rnd(rnd(q) * s)
    if W_group_mode == 3:
        if zero_is_scalar:  # integer subtract first, then cast and scale
            return rnd(rnd(q - z) * s)
        return rnd(rnd(rnd(q) - z) * s)
    if W_group_mode == 4:  # fma(b.to(meta), scales, zeros): ONE rounding
        return rnd(rnd(q) * s + z)
    raise ValueError(W_group_mode)


def forward(x, W_kn: np.ndarray, *, scales_w_channel=None, scales_x=None, channel_scale_mode=0,
            bias=None, meta_code=None, output_code=None) -> np.ndarray:
    """y[M,N] = epilogue(x[M,K] @ W[K,N]) in float64.

    Epilogue order (gemm_kernels.py:392-406, core.py:192-193): K-reduction -> channel scaling
    (in meta dtype when meta_code is given) -> cast to output dtype -> + bias.
    """
    acc = to_f64(x) @ W_kn
    rnd = (lambda a: a) if meta_code is None else (lambda a: round_to_dtype(a, meta_code))
    if channel_scale_mode == 1:
        acc = rnd(acc) * np.asarray(scales_w_channel, np.float64).reshape(1, -1)
    elif channel_scale_mode == 2:
        acc = rnd(acc) * np.asarray(scales_x, np.float64).reshape(-1, 1)
    elif channel_scale_mode == 3:
        acc = rnd(acc) * (np.asarray(scales_x, np.float64).reshape(-1, 1)
                          * np.asarray(scales_w_channel, np.float64).reshape(1, -1))
    if output_code is not None:
        acc = round_to_dtype(acc, output_code)
    if bias is not None:
        acc = acc + to_f64(bias).reshape(1, -1)
        if output_code is not None:
            acc = round_to_dtype(acc, output_code)
    return acc


def forward_packed(x, packed, scales_gn, zeros_gn, *, W_nbits, group_size, W_group_mode,
                   channel_scale_mode=0, scales_x=None, zero_is_scalar=False, pack_bits=32,
                   bias=None, meta_code=None, output_code=None, weight_cast_code=None) -> np.ndarray:
    """End-to-end oracle for packed weights: unpack -> dequantize -> matmul -> epilogue.
    weight_cast_code: the reference casts the dequantised tile to the INPUT dtype before the dot
    (`b.to(input_dtype)`, gemm_kernels.py:384) — a real rounding step when the activations are fp8."""
    q_kn = unpack_over_cols(np.asarray(packed), W_nbits, pack_bits).T.astype(np.float64)
    grouped = W_group_mode in (2, 3, 4) or (W_group_mode == 1 and channel_scale_mode not in (1, 3))
    W = dequantize(q_kn, scales_gn if W_group_mode >= 2 else None,
                   zeros_gn if W_group_mode in (1, 3, 4) else None,
                   group_size, W_group_mode, zero_is_scalar, meta_code)
    del grouped
    if weight_cast_code is not None:
        W = round_to_dtype(W, weight_cast_code)
    ch = scales_gn if channel_scale_mode in (1, 3) else None
    return forward(x, W, scales_w_channel=ch, scales_x=scales_x, channel_scale_mode=channel_scale_mode,
                   bias=bias, meta_code=meta_code, output_code=output_code)


# --------------------------------------------------------------------------------------
# per-token dynamic activation quantisation
# (spec: gemlite/quant_utils.py:231-253; AMD rounding floor(x+0.5): :259-266, :286-298)
# --------------------------------------------------------------------------------------
def scale_activations_per_token(x, out_code: int):
    """x[M,K] -> (x_q[M,K] as float64 values on the int8/fp8 grid, scales[M,1] float32)."""
    xf = to_f64(x).astype(np.float32)  # fp32_scale=True path
    qmin, qmax = {INT8: (-128.0, 127.0), FP8E4: (-448.0, 448.0), FP8E5: (-57344.0, 57344.0)}[out_code]
    s = np.abs(xf).max(axis=1, keepdims=True).astype(np.float32)
    s = (s / np.float32(qmax)).astype(np.float32)
    s = np.maximum(s, np.float32(1e-6))
    y = (xf / s).astype(np.float32)
    y = np.clip(y, np.float32(qmin), np.float32(qmax))
    if out_code == INT8:
        y = np.floor(y + np.float32(0.5))  # the reference's AMD rounding
        y = np.clip(y, -128, 127)
        return y.astype(np.float64), s
    return round_to_dtype(y.astype(np.float64), out_code), s


# --------------------------------------------------------------------------------------
# synthetic data of SURVEY.md §8(d) / tests/test_gemlitelineartriton.py:25-45 (seeded)
# --------------------------------------------------------------------------------------
def gen_data(N: int, K: int, W_nbits: int, group_size: int, seed: int = 0, np_float=np.float16):
    rng = np.random.default_rng(seed)
    W_q = rng.integers(0, 2 ** W_nbits, size=(N, K), dtype=np.uint8)
    ng = N * K // group_size
    scales = (rng.random((ng, 1), dtype=np.float32) * 0.01 + 0.001).astype(np_float)
    zeros = (rng.random((ng, 1), dtype=np.float32) * (2 ** W_nbits - 1)).astype(np_float)
    return W_q, scales, zeros


def gen_x(M: int, K: int, seed: int = 1, np_float=np.float16):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((M, K), dtype=np.float32) / 10.0).astype(np_float)
