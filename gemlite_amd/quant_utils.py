"""Dynamic per-token activation quantisation (reference: gemlite/quant_utils.py:231-347).

``scale_activations_per_token(x, w_dtype)`` returns ``(x_q, scales)`` with ``x_q`` of dtype ``w_dtype``
(int8 / fp8) and ``scales`` fp32 ``[M, 1]``:  s = max(amax|x_row| / qmax, 1e-6), x_q = clamp(x / s),
int8 rounded with floor(v + 0.5) — the rounding the reference's kernel uses on AMD (quant_utils.py:259-266).
Runs as one HIP kernel (`gemlite_hip_scale_activations_per_token`).  MXFP / NVFP activation formats are
out of scope.
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
    rc = _hip.load().gemlite_hip_scale_activations_per_token(
        x2.data_ptr(), y.data_ptr(), scales.data_ptr(), M, K, x2.stride(0), TORCH_TO_DTYPE[x2.dtype].value,
        TORCH_TO_DTYPE[w_dtype].value, _hip.current_stream_handle(tensor.device))
    _hip.raise_for_status(rc, "scale_activations_per_token")
    if not fp32_scale:
        scales = scales.to(tensor.dtype)
    return y.view(shape), scales


scale_activations_per_token_triton = scale_activations_per_token  # reference export name


def _mx_unsupported(*_a, **_k):
    raise NotImplementedError("MXFP / NVFP activation formats are outside this build's scope (SURVEY.md §8 a)")


scale_activations_mxfp8 = scale_activations_mxfp4 = scale_activations_nvfp4 = _mx_unsupported
