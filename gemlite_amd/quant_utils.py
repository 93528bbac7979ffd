"""Dynamic per-token activation quantisation (reference: gemlite/quant_utils.py:231-347).

``scale_activations_per_token(x, w_dtype)`` returns ``(x_q, scales)`` with ``x_q`` of dtype ``w_dtype``
(int8 / fp8) and ``scales`` fp32 ``[M, 1]``:  s = max(amax|x_row| / qmax, 1e-6), x_q = clamp(x / s),
int8 rounded with floor(v + 0.5) — the rounding the reference's kernel uses on AMD (quant_utils.py:259-266).
Runs as one HIP kernel (`gemlite_hip_scale_activations_per_token`).

Block-scaled formats (reference: gemlite/quant_utils.py:21-225 weight quantiser, :502-954 activation quantisers):
``WeightQuantizerMXFP`` (host-side torch code, any device: it runs once per layer) and
``scale_activations_mxfp8 / _mxfp4 / _nvfp4`` (one HIP kernel each, `gemlite_hip_scale_activations_*`).
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _hip
from .dtypes import TORCH_TO_DTYPE


def get_dtype_range(dtype: torch.dtype) -> Tuple[float, float]:
    if dtype.is_floating_point:
        info = torch.finfo(dtype)
    else:
        info = torch.iinfo(dtype)
    return float(info.min), float(info.max)


def scale_activations_per_token(tensor: torch.Tensor, w_dtype: torch.dtype, fp32_scale: bool = True):
    _hip.require_gpu_tensor(tensor, "tensor")
    if w_dtype not in (torch.int8, torch.float8_e4m3fn, torch.float8_e5m2):
        raise NotImplementedError(f"activation quantisation to {w_dtype} is not supported on gfx950 "
                                  "(use torch.float8_e4m3fn, not the MI300X fnuz flavour)")
    shape = tensor.shape
    x2 = tensor.reshape(-1, shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M, K = x2.shape
    y = torch.empty((M, K), dtype=w_dtype, device=tensor.device)
    scales = torch.empty((M, 1), dtype=torch.float32, device=tensor.device)
    with _hip.on_device(tensor.device):
        rc = _hip.load().gemlite_hip_scale_activations_per_token(
            x2.data_ptr(), y.data_ptr(), scales.data_ptr(), M, K, x2.stride(0), TORCH_TO_DTYPE[x2.dtype].value,
            TORCH_TO_DTYPE[w_dtype].value, _hip.current_stream_handle(tensor.device))
    _hip.raise_for_status(rc, "scale_activations_per_token")
    if not fp32_scale:
        scales = scales.to(tensor.dtype)
    return y.view(shape), scales


scale_activations_per_token_triton = scale_activations_per_token  # reference export name


# ------------------------------------------------------------------------------------------------------
# block-scaled formats
# ------------------------------------------------------------------------------------------------------
NVFP4_META_SCALE = 0.05  # the reference's fixed second-level scale of NVFP4 (quant_utils.py:21)
MX_EPS_EXP = -30         # smallest block scale: 2^-30
FP4_POS_VALUES = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)          # e2m1 magnitudes, code = index (+8: negative)
FP4_THRESHOLDS = (0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0)          # midpoints: a magnitude ON a midpoint rounds down


def _fp4_tables(device, dtype=torch.float32):
    pos = torch.tensor(FP4_POS_VALUES, dtype=dtype, device=device)
    return pos, torch.tensor(FP4_THRESHOLDS, dtype=dtype, device=device), torch.cat([pos, -pos])


class WeightQuantizerMXFP:
    """Weights -> (elements, block scales) in the OCP microscaling formats, arithmetic of the reference quantiser
    (quant_utils.py:70-225): MXFP8 / MXFP4 = e4m3 / e2m1 elements with one power-of-two (e8m0) scale per 32 weights,
    ``scale = 2^ceil(log2(amax / qmax))`` floored at 2^-30; NVFP4 = e2m1 elements with an e4m3 scale per 16 weights on
    top of the fixed meta scale 0.05.  ``index=True`` returns what ``GemLiteLinear.pack`` takes (fp8 tensor / uint8
    codes), otherwise the rounded values as floats.  ``window_size`` > 0 searches neighbouring scales for the smallest
    mean absolute error, like the reference."""

    def __init__(self, compute_dtype=torch.bfloat16, device="cuda:0"):
        self.compute_dtype = compute_dtype
        self.device = device

    @staticmethod
    def round_to_closest_fp4(tensor: torch.Tensor) -> torch.Tensor:
        pos, thr, _ = _fp4_tables(tensor.device)
        out = pos[torch.searchsorted(thr.to(tensor.dtype), tensor.abs())].to(tensor.dtype)
        return out * tensor.sign()

    @staticmethod
    def to_index(W_q: torch.Tensor) -> torch.Tensor:
        assert W_q.is_floating_point(), "Input should be floating point fp4 values."
        _, _, values = _fp4_tables(W_q.device, W_q.dtype)
        hit = W_q.reshape(-1, 1) == values.view(1, -1)  # -0.0 == 0.0: both zeros map to code 0 (first match)
        return hit.to(torch.uint8).argmax(dim=1).to(torch.uint8).view(W_q.shape)

    def quantize_mxfp8(self, W, index: bool = False, mx_fp8_dtype: torch.dtype = torch.float8_e4m3fn):
        eps = 2.0 ** MX_EPS_EXP
        lo, hi = get_dtype_range(mx_fp8_dtype)
        flat = W.reshape(-1, 32).float()
        ideal = flat.abs().amax(dim=1, keepdim=True) / hi
        scales = (2 ** torch.ceil(torch.log2(ideal))).clamp_(min=eps)
        W_q = (flat / scales).clamp_(min=lo, max=hi).to(mx_fp8_dtype)
        if not index:
            W_q = W_q.to(flat.dtype)
        return W_q, scales.to(torch.float8_e8m0fnu)

    def _search(self, flat, candidates, full_scale_of):
        """candidate with the smallest mean |W - round(W / s) * s| per block"""
        q = self.round_to_closest_fp4(flat.unsqueeze(1) / full_scale_of(candidates).unsqueeze(-1))
        err = (flat.unsqueeze(1) - q * candidates.unsqueeze(-1)).abs().mean(dim=-1)
        return torch.gather(candidates, 1, torch.argmin(err, dim=1, keepdim=True))

    def quantize_mxfp4(self, W, window_size: int = 0, index: bool = False):
        eps = 2.0 ** MX_EPS_EXP
        flat = W.reshape(-1, 32).float()
        ideal = flat.abs().amax(dim=1, keepdim=True) / 6
        log2s = torch.ceil(torch.log2(ideal))
        if window_size == 0:
            scales = 2 ** log2s
        else:
            offs = torch.arange(-window_size, window_size + 1, device=W.device, dtype=log2s.dtype).view(1, -1)
            cand = torch.pow(2, log2s + offs)
            cand[cand < eps] = eps
            scales = self._search(flat, cand, lambda c: c)
        scales = scales.clamp_(eps)
        W_q = self.round_to_closest_fp4(flat / scales)
        if index:
            W_q = self.to_index(W_q)
        return W_q, scales.to(torch.float8_e8m0fnu)

    def quantize_nvfp4(self, W, window_size: int = 0, index: bool = False):
        eps, fp8 = 1e-6, torch.float8_e4m3fn
        flat = W.reshape(-1, 16).float()
        ideal = flat.abs().amax(dim=1, keepdim=True) / 6
        ideal = (ideal / NVFP4_META_SCALE).clamp_(max=torch.finfo(fp8).max).to(fp8)
        if window_size == 0:
            scales = ideal
        else:
            offs = torch.arange(-window_size, window_size + 1, device=W.device, dtype=torch.int).view(1, -1)
            cand = (ideal.view(torch.int8) + offs).clamp_(-128, 127).to(torch.int8)
            cand[cand == -1] = 1   # the two NaN encodings of e4m3
            cand[cand == 127] = 1
            cand = cand.view(fp8).float()
            cand[cand < eps] = eps
            q = self.round_to_closest_fp4(flat.unsqueeze(1) / (cand * NVFP4_META_SCALE).unsqueeze(-1))
            err = (flat.unsqueeze(1) - q * cand.unsqueeze(-1)).abs().mean(dim=-1)
            scales = torch.gather(cand, 1, torch.argmin(err, dim=1, keepdim=True)).to(fp8)
        full = (scales.to(flat.dtype) * NVFP4_META_SCALE).clamp_(min=eps)
        W_q = self.round_to_closest_fp4(flat / full)
        if index:
            W_q = self.to_index(W_q)
        return W_q, scales

    def dequantize(self, W_q, scales, shape=None, dtype=None):
        if W_q.dtype == torch.uint8:  # e2m1 codes
            _, _, values = _fp4_tables(W_q.device)
            W_q = values[W_q.int()]
        group = W_q.numel() // scales.numel()
        out = W_q.reshape(-1, group).float() * scales.float().reshape(-1, 1)
        if shape is not None:
            out = out.view(shape)
        return out.to(self.compute_dtype if dtype is None else dtype)


def _scale_activations_mx(tensor: torch.Tensor, mode: str):
    """(x_q, block scales) of the activations.  x_q: fp8 [.., K] (mxfp8) or uint8 [.., K/2] e2m1 codes, k even in the low
    nibble (mxfp4 / nvfp4); scales: uint8 e8m0 [M_pad, K/32] (nvfp4: float8_e4m3fn [M_pad, K/16]) with M_pad = M rounded up
    to the block size — shapes and padding of scale_activations_*_triton_v2 (quant_utils.py:546-590, 820-855, 917-954)."""
    _hip.require_gpu_tensor(tensor, "tensor")
    group = 16 if mode == "nvfp4" else 32
    x2 = tensor.reshape(-1, tensor.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M, K = x2.shape
    if K % 32 != 0:
        raise NotImplementedError(f"block-scaled activation quantisation needs K % 32 == 0, got K = {K}")
    m_pad = (M + group - 1) // group * group
    lib = _hip.load()
    if mode == "mxfp8":
        y = torch.empty((M, K), dtype=torch.float8_e4m3fn, device=tensor.device)
        fn = lib.gemlite_hip_scale_activations_mxfp8
    else:
        y = torch.empty((M, K // 2), dtype=torch.uint8, device=tensor.device)
        fn = lib.gemlite_hip_scale_activations_mxfp4 if mode == "mxfp4" else lib.gemlite_hip_scale_activations_nvfp4
    scales = torch.empty((m_pad, K // group), dtype=torch.uint8, device=tensor.device)
    with _hip.on_device(tensor.device):
        rc = fn(x2.data_ptr(), y.data_ptr(), scales.data_ptr(), M, K, x2.stride(0), TORCH_TO_DTYPE[x2.dtype].value,
                _hip.current_stream_handle(tensor.device))
    _hip.raise_for_status(rc, "scale_activations_" + mode)
    if mode == "nvfp4":
        scales = scales.view(torch.float8_e4m3fn)
    return y, scales


def scale_activations_mxfp8(tensor: torch.Tensor, w_dtype: torch.dtype = torch.float8_e4m3fn):
    if w_dtype != torch.float8_e4m3fn:
        raise NotImplementedError("MXFP8 activations are OCP e4m3 on gfx950")
    return _scale_activations_mx(tensor, "mxfp8")


def scale_activations_mxfp4(tensor: torch.Tensor):
    return _scale_activations_mx(tensor, "mxfp4")


def scale_activations_nvfp4(tensor: torch.Tensor):
    return _scale_activations_mx(tensor, "nvfp4")


# the reference's export names for the same entry points
scale_activations_mxfp8_triton_v2, scale_activations_mxfp4_triton_v2 = scale_activations_mxfp8, scale_activations_mxfp4
scale_activations_nvfp4_triton_v2 = scale_activations_nvfp4
