"""Bit-packing of quantised weights — same bit layout as the reference, different machinery.

Layout contract (reference: gemlite/bitpack.py:36-60 `pack_weights_over_cols_torch`, :10-34 rows variant;
consumers: gemm_kernels.py:327-328): for ``e = packing_bitwidth // W_nbits`` the word for output column
``j`` of row ``n`` is ``OR_i W_q[n, j*e+i] << (W_nbits*i)`` (element i at bits [b*i, b*i+b), LSB first),
stored in ``uint8 / int16 / int32 / int64``.  ``GemLiteLinear.pack`` asks for ``transpose=True`` and makes
the result contiguous: ``[K/e, N]`` with N fastest — the layout every HIP kernel streams.

On a GPU tensor the pack / unpack runs as a HIP kernel (``gemlite_hip_pack_over_cols``); on a CPU tensor
it is a vectorised shift-and-sum (one pass, no Python loop over K) — host-side model loading, not the
forward hot path.
"""
from __future__ import annotations

import torch

from . import _hip
from .dtypes import PACKING_BITWIDTH_TO_TORCH_DTYPE

_SUPPORTED_PACK = (8, 16, 32, 64)
_SUPPORTED_BITS = (8, 4, 2, 1)


def _check(W_nbits: int, packing_bitwidth: int):
    assert packing_bitwidth in _SUPPORTED_PACK, "Unsuported bitpacking width"
    assert W_nbits in _SUPPORTED_BITS, "Unsuported nbits"
    return packing_bitwidth // W_nbits


def _pack_lastdim_cpu(W_q: torch.Tensor, W_nbits: int, packing_bitwidth: int) -> torch.Tensor:
    """[R, C] values -> [R, C/e] words (vectorised)."""
    e = packing_bitwidth // W_nbits
    R, Cc = W_q.shape
    assert Cc % e == 0, "the packed dimension must be a multiple of elements_per_sample"
    w = W_q.to(torch.int64).reshape(R, Cc // e, e)
    shifts = torch.arange(e, dtype=torch.int64, device=W_q.device) * W_nbits
    words = (w << shifts).sum(dim=-1)  # disjoint bit fields: sum == OR (wraps like a C cast at 64 bits)
    if packing_bitwidth == 8:
        return words.to(torch.uint8)
    if packing_bitwidth < 64:
        half = 1 << (packing_bitwidth - 1)  # wrap into the signed storage type
        words = ((words + half) % (1 << packing_bitwidth)) - half
    return words.to(PACKING_BITWIDTH_TO_TORCH_DTYPE[packing_bitwidth])


def _unpack_lastdim_cpu(W_p: torch.Tensor, W_nbits: int, packing_bitwidth: int) -> torch.Tensor:
    e = packing_bitwidth // W_nbits
    w = W_p.to(torch.int64).unsqueeze(-1)
    shifts = torch.arange(e, dtype=torch.int64, device=W_p.device) * W_nbits
    out = (w >> shifts) & ((1 << W_nbits) - 1)
    return out.reshape(W_p.shape[0], W_p.shape[1] * e).to(torch.uint8)


def pack_weights_over_cols(W_q: torch.Tensor, W_nbits: int, packing_bitwidth: int, transpose: bool):
    """Pack along the last dim of ``W_q[N, K]`` (K for a linear layer).  Returns ``(packed, e)``;
    ``packed`` is ``[K/e, N]`` when ``transpose`` else ``[N, K/e]``."""
    e = _check(W_nbits, packing_bitwidth)
    assert W_q.dim() == 2
    if W_q.is_cuda:
        N, K = W_q.shape
        assert K % e == 0
        src = W_q if (W_q.dtype == torch.uint8 and W_q.stride(1) == 1) else W_q.to(torch.uint8).contiguous()
        out = torch.empty((K // e, N), dtype=PACKING_BITWIDTH_TO_TORCH_DTYPE[packing_bitwidth], device=W_q.device)
        with _hip.on_device(W_q.device):
            rc = _hip.load().gemlite_hip_pack_over_cols(src.data_ptr(), out.data_ptr(), N, K, src.stride(0), W_nbits,
                                                        packing_bitwidth, _hip.current_stream_handle(W_q.device))
        _hip.raise_for_status(rc, "pack_over_cols")
        return (out if transpose else out.t().contiguous()), e
    packed = _pack_lastdim_cpu(W_q, W_nbits, packing_bitwidth)
    return (packed.t() if transpose else packed), e


def unpack_over_cols(W_q_packed: torch.Tensor, W_nbits: int, num_output_cols: int, dtype: torch.dtype = torch.uint8):
    """Inverse of ``pack_weights_over_cols(..., transpose=False)``: ``[N, K/e]`` words -> ``[N, K]``."""
    pb = W_q_packed.element_size() * 8
    e = _check(W_nbits, pb)
    assert num_output_cols == W_q_packed.shape[1] * e
    if W_q_packed.is_cuda:
        N, K = W_q_packed.shape[0], num_output_cols
        src = W_q_packed.t().contiguous()  # the kernel reads the [K/e, N] layout
        out = torch.empty((N, K), dtype=torch.uint8, device=W_q_packed.device)
        with _hip.on_device(W_q_packed.device):
            rc = _hip.load().gemlite_hip_unpack_over_cols(src.data_ptr(), out.data_ptr(), N, K, W_nbits, pb,
                                                          _hip.current_stream_handle(W_q_packed.device))
        _hip.raise_for_status(rc, "unpack_over_cols")
        return out.to(dtype)
    return _unpack_lastdim_cpu(W_q_packed, W_nbits, pb).to(dtype)


def pack_weights_over_rows(W_q: torch.Tensor, W_nbits: int, packing_bitwidth: int, transpose: bool):
    """Pack along dim 0 (reference: bitpack.py:10-34).  Model-loading utility, torch ops only."""
    e = _check(W_nbits, packing_bitwidth)
    packed = _pack_lastdim_cpu(W_q.t().contiguous().cpu(), W_nbits, packing_bitwidth).t().contiguous().to(W_q.device)
    return (packed.t() if transpose else packed), e


def unpack_over_rows(W_q_packed: torch.Tensor, W_nbits: int, num_output_rows: int, dtype: torch.dtype = torch.uint8):
    pb = W_q_packed.element_size() * 8
    e = _check(W_nbits, pb)
    assert num_output_rows == W_q_packed.shape[0] * e
    out = _unpack_lastdim_cpu(W_q_packed.t().contiguous().cpu(), W_nbits, pb).t().contiguous()
    return out.to(device=W_q_packed.device, dtype=dtype)


# names the reference exports (bitpack.py) — kept so call sites can switch packages unchanged
pack_weights_over_cols_torch = pack_weights_over_cols
pack_weights_over_cols_triton = pack_weights_over_cols  # "triton" == "the GPU one": HIP here
pack_weights_over_rows_torch = pack_weights_over_rows
pack_weights_over_rows_triton = pack_weights_over_rows
unpack_over_cols_torch = unpack_over_cols
unpack_over_cols_triton = unpack_over_cols
unpack_over_rows_torch = unpack_over_rows
unpack_over_rows_triton = unpack_over_rows
