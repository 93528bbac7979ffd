"""CPU ORACLE for the block-scaled formats (MXFP8 / MXFP4 / NVFP4) — TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's block-scaled path.  Nothing in ``gemlite_amd/`` may import this file: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker.

Parity pinning (tests/test_oracle_golden.py::test_mx_*): the weight quantisers, the three activation quantisers and the
processors' ``pack()`` output are checked bit for bit against ``tests/golden/mx.npz`` = outputs of THE REFERENCE ITSELF
(generator: oracle/gen_golden_mx.py; torch code on CPU, Triton kernels under TRITON_INTERPRET=1).  The MX matmul kernels
(``tl.dot_scaled``) do not run under the CPU interpreter, so ``mx_matmul`` is pinned by DEFINITION only: dot_scaled is
"dequantise both operands block-wise, multiply, accumulate in fp32" (gemm_kernels.py:505-531), which is what it computes,
plus the reference's own acceptance bar (tests/test_mxfp.py: mean |y - linear(x)| below 2e-4 .. 1e-3).

Each function cites the reference file:line it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import numpy as np

try:  # torch only to move bf16 / fp8 bit patterns in and out of float32
    import torch
except Exception:  # pragma: no cover
    torch = None

NVFP4_META_SCALE = 0.05                                                   # quant_utils.py:21
FP4_VALUES = np.array([0, 0.5, 1, 1.5, 2, 3, 4, 6, -0.0, -0.5, -1, -1.5, -2, -3, -4, -6], dtype=np.float32)  # :31-37
FP4_THRESHOLDS = np.array([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0], dtype=np.float32)                          # :47-53
THR_POS = np.array([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 7.0], dtype=np.float32)                            # :61-68


# --------------------------------------------------------------------------------------
# element formats
# --------------------------------------------------------------------------------------
def fp8_e4m3_decode(b: np.ndarray) -> np.ndarray:
    """OCP e4m3fn bytes -> float32 (bias 7, no inf, 0x7F / 0xFF = NaN)."""
    b = np.asarray(b, dtype=np.uint8).astype(np.int32)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    mag = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * np.exp2((e - 7).astype(np.float64)))
    mag = np.where((e == 15) & (m == 7), np.nan, mag)
    return np.where(s == 1, -mag, mag).astype(np.float32)


def fp8_e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float32 -> e4m3fn bytes, round to nearest even (what `.to(float8_e4m3fn)` / `.to(tl.float8e4nv)` do)."""
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.float8_e4m3fn)
    return t.view(torch.uint8).numpy()


def e8m0_decode(b: np.ndarray) -> np.ndarray:
    """e8m0 bytes -> float64 2^(b - 127)."""
    return np.exp2(np.asarray(b, dtype=np.uint8).astype(np.float64) - 127.0)


def fp4_unpack(packed: np.ndarray) -> np.ndarray:
    """uint8 [.., K/2] (k even in the low nibble, bitpack.py:36-60 with 8-bit words) -> float32 e2m1 values [.., K]."""
    p = np.asarray(packed, dtype=np.uint8)
    codes = np.stack([p & 15, p >> 4], axis=-1).reshape(*p.shape[:-1], p.shape[-1] * 2)
    return FP4_VALUES[codes]


def fp4_pack_codes(codes: np.ndarray) -> np.ndarray:
    c = np.asarray(codes, dtype=np.uint8)
    return (c[..., 0::2] | (c[..., 1::2] << 4)).astype(np.uint8)


# --------------------------------------------------------------------------------------
# weight quantiser (quant_utils.py:70-225); W float32 [N, K]
# --------------------------------------------------------------------------------------
def _f32(a):
    return np.asarray(a, dtype=np.float32)


def round_to_closest_fp4(t: np.ndarray) -> np.ndarray:
    """quant_utils.py:75-84: searchsorted(thresholds, |t|) (a magnitude ON a midpoint takes the lower value), * sign."""
    t = _f32(t)
    idx = np.searchsorted(FP4_THRESHOLDS, np.abs(t), side="left")
    return (FP4_VALUES[:8][idx] * np.sign(t)).astype(np.float32)


def fp4_to_index(v: np.ndarray) -> np.ndarray:
    """quant_utils.py:86-95: first code whose value compares equal (-0.0 == 0.0 -> code 0)."""
    v = _f32(v)
    return (v.reshape(-1, 1) == FP4_VALUES.reshape(1, -1)).argmax(axis=1).astype(np.uint8).reshape(v.shape)


def _pow2_ceil_scale(ideal: np.ndarray, eps: float) -> np.ndarray:
    with np.errstate(divide="ignore"):
        s = np.exp2(np.ceil(np.log2(ideal.astype(np.float32)))).astype(np.float32)
    return np.maximum(s, np.float32(eps))


def _to_e8m0(scales_f32: np.ndarray) -> np.ndarray:
    """powers of two in [2^-30, 2^127] -> exponent bytes (what `.to(float8_e8m0fnu).view(uint8)` yields for them)."""
    return ((scales_f32.astype(np.float32).view(np.uint32) >> 23) & 0xFF).astype(np.uint8)


def quantize_mxfp8(W: np.ndarray):
    """quant_utils.py:97-124 with e4m3fn: -> (fp8 bytes [N*K/32, 32], e8m0 bytes [N*K/32, 1])."""
    flat = _f32(W).reshape(-1, 32)
    ideal = np.abs(flat).max(axis=1, keepdims=True) / np.float32(448.0)
    scales = _pow2_ceil_scale(ideal, 2.0 ** -30)
    q = np.clip(flat / scales, -448.0, 448.0).astype(np.float32)
    return fp8_e4m3_encode(q), _to_e8m0(scales)


def quantize_mxfp4(W: np.ndarray, window_size: int = 0):
    """quant_utils.py:126-163 -> (e2m1 codes [N*K/32, 32], e8m0 bytes)."""
    eps = np.float32(2.0 ** -30)
    flat = _f32(W).reshape(-1, 32)
    ideal = np.abs(flat).max(axis=1, keepdims=True) / np.float32(6)
    with np.errstate(divide="ignore"):
        log2s = np.ceil(np.log2(ideal.astype(np.float32))).astype(np.float32)
    if window_size == 0:
        scales = np.exp2(log2s).astype(np.float32)
    else:
        offs = np.arange(-window_size, window_size + 1, dtype=np.float32).reshape(1, -1)
        cand = np.power(np.float32(2), log2s + offs).astype(np.float32)
        cand[cand < eps] = eps
        q = round_to_closest_fp4(flat[:, None, :] / cand[:, :, None])
        err = np.abs(flat[:, None, :] - q * cand[:, :, None]).astype(np.float32).mean(axis=-1, dtype=np.float32)
        scales = np.take_along_axis(cand, err.argmin(axis=1)[:, None], axis=1)
    scales = np.maximum(scales, eps)
    return fp4_to_index(round_to_closest_fp4(flat / scales)), _to_e8m0(scales)


def quantize_nvfp4(W: np.ndarray, window_size: int = 0):
    """quant_utils.py:165-213 -> (e2m1 codes [N*K/16, 16], e4m3 bytes)."""
    eps = np.float32(1e-6)
    meta = np.float32(NVFP4_META_SCALE)
    flat = _f32(W).reshape(-1, 16)
    ideal = np.abs(flat).max(axis=1, keepdims=True) / np.float32(6)
    ideal = np.minimum(ideal / meta, np.float32(448.0)).astype(np.float32)
    s8 = fp8_e4m3_encode(ideal)
    if window_size > 0:
        offs = np.arange(-window_size, window_size + 1, dtype=np.int32).reshape(1, -1)
        cand = np.clip(s8.view(np.int8).astype(np.int32) + offs, -128, 127).astype(np.int8)
        cand[cand == -1] = 1
        cand[cand == 127] = 1
        cand = fp8_e4m3_decode(cand.view(np.uint8))
        cand[cand < eps] = eps
        q = round_to_closest_fp4(flat[:, None, :] / (cand * meta)[:, :, None])
        err = np.abs(flat[:, None, :] - q * cand[:, :, None]).astype(np.float32).mean(axis=-1, dtype=np.float32)
        s8 = fp8_e4m3_encode(np.take_along_axis(cand, err.argmin(axis=1)[:, None], axis=1))
    full = np.maximum(fp8_e4m3_decode(s8) * meta, eps).astype(np.float32)
    return fp4_to_index(round_to_closest_fp4(flat / full)), s8


# --------------------------------------------------------------------------------------
# activation quantisers (the *_triton_v2 kernels); x float32 [M, K] -> (elements, scales [M_pad, K/g])
# --------------------------------------------------------------------------------------
def _next_pow2_bitwise(val: np.ndarray, eps_exp: int = -30):
    """quant_utils.py:381-392 next_power_of_2_bitwise_triton: exponent field (+1 when the mantissa is not zero)."""
    xi = np.ascontiguousarray(val, dtype=np.float32).view(np.uint32)
    ex = ((xi >> 23) & 0xFF).astype(np.int32) + ((xi & 0x7FFFFF) != 0)
    ex = np.maximum(np.minimum(ex, 254), 127 + eps_exp)
    return (ex.astype(np.uint32) << 23).view(np.float32), ex.astype(np.uint8)


def _pad_rows(x: np.ndarray, group: int) -> np.ndarray:
    M = x.shape[0]
    pad = (group - M % group) % group
    return np.concatenate([x, np.zeros((pad, x.shape[1]), dtype=x.dtype)], axis=0) if pad else x


def _fp4_codes(wq: np.ndarray) -> np.ndarray:
    """quant_utils.py:800-802: count of thresholds below |wq|, + 8 unless wq >= 0."""
    idx = (np.abs(wq)[..., None] > THR_POS).sum(axis=-1).astype(np.uint8)
    return np.where(wq >= 0, idx, idx + 8).astype(np.uint8)


def scale_activations_mxfp8(x: np.ndarray):
    """quant_utils.py:502-590 -> (fp8 bytes [M, K], e8m0 bytes [M_pad, K/32])."""
    x = _f32(x)
    M, K = x.shape
    xp = _pad_rows(x, 32).reshape(-1, K // 32, 32)
    scales, ex = _next_pow2_bitwise(np.abs(xp).max(axis=-1) / np.float32(448.0))
    q = np.clip(xp / scales[..., None], -448.0, 448.0).astype(np.float32)
    return fp8_e4m3_encode(q.reshape(-1, K)[:M]), ex


def scale_activations_mxfp4(x: np.ndarray):
    """quant_utils.py:769-855 -> (uint8 [M, K/2], e8m0 bytes [M_pad, K/32])."""
    x = _f32(x)
    M, K = x.shape
    xp = _pad_rows(x, 32).reshape(-1, K // 32, 32)
    scales, ex = _next_pow2_bitwise(np.abs(xp).max(axis=-1) / np.float32(6.0))
    codes = _fp4_codes((xp / scales[..., None]).astype(np.float32)).reshape(-1, K)[:M]
    return fp4_pack_codes(codes), ex


def scale_activations_nvfp4(x: np.ndarray):
    """quant_utils.py:859-954 -> (uint8 [M, K/2], e4m3 bytes [M_pad, K/16])."""
    x = _f32(x)
    M, K = x.shape
    xp = _pad_rows(x, 16).reshape(-1, K // 16, 16)
    s = np.abs(xp).max(axis=-1) / np.float32(6.0 * NVFP4_META_SCALE)
    s8 = fp8_e4m3_encode(np.minimum(s, np.float32(448.0)).astype(np.float32))
    full = np.maximum(fp8_e4m3_decode(s8) * np.float32(NVFP4_META_SCALE), np.float32(1e-6)).astype(np.float32)
    codes = _fp4_codes((xp / full[..., None]).astype(np.float32)).reshape(-1, K)[:M]
    return fp4_pack_codes(codes), s8


# --------------------------------------------------------------------------------------
# matmul: tl.dot_scaled semantics (gemm_kernels.py:505-531, 534-546)
# --------------------------------------------------------------------------------------
def dequant_blocks(elems_f32: np.ndarray, scale_bytes: np.ndarray, group: int, e4m3_scales: bool = False) -> np.ndarray:
    """[R, K] element values * their [R, K/group] block scales -> float64 [R, K]."""
    s = fp8_e4m3_decode(scale_bytes).astype(np.float64) if e4m3_scales else e8m0_decode(scale_bytes)
    R, K = elems_f32.shape
    return (elems_f32.astype(np.float64).reshape(R, K // group, group) * s[:R, :, None]).reshape(R, K)


def mx_matmul(x_vals: np.ndarray, w_vals_nk: np.ndarray, *, sx=None, sw=None, group: int = 32, e4m3_scales: bool = False,
              scales_x_token=None, post: float = 1.0) -> np.ndarray:
    """out[M, N] = post * sum_k (x[m, k] sx[m, k/g]) (w[n, k] sw[n, k/g])  [* scales_x_token[m]]   in float64.

    x_vals / w_vals_nk: element VALUES (float32; fp8 / fp4 already decoded, 16-bit activations as they are);
    sx / sw: block-scale bytes [M_pad, K/g] / [N, K/g] or None."""
    xd = x_vals.astype(np.float64) if sx is None else dequant_blocks(x_vals, sx, group, e4m3_scales)
    wd = w_vals_nk.astype(np.float64) if sw is None else dequant_blocks(w_vals_nk, sw, group, e4m3_scales)
    out = xd @ wd.T
    if scales_x_token is not None:
        out = out * np.asarray(scales_x_token, dtype=np.float64).reshape(-1, 1)
    return out * post
