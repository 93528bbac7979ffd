"""Layer "processors": quantise / wrap weights into a packed GemLiteLinear (reference: gemlite/helper.py).

Round-1 scope: the processors whose kernels are on the north-star path — A16W8 / A16Wn from raw
(W_q, scales, zeros), A8W8 dynamic (int8 / fp8) — plus ``patch_model`` for ``torch.nn.Linear`` layers.
HQQ-object inputs (``from_hqqlinear``) need the third-party ``hqq`` package, which is not installed in this
image; the raw-tensor entry points take exactly what ``HQQLinear.unpack()`` / ``meta`` hold.
On gfx950 FP8 means OCP e4m3fn (the reference's HIP default e4m3fnuz, helper.py:13-15, is the MI300X format).
"""
from typing import Optional

import torch

from .core import GemLiteLinear
from .dtypes import TORCH_TO_DTYPE, DType

default_fp8 = torch.float8_e4m3fn
default_post_scale = True  # channel-wise scaling applied after the K reduction (reference HIP default)


def _gemlite_dtype(dtype: torch.dtype) -> DType:
    assert dtype in (torch.float16, torch.bfloat16), "compute dtype should be float16 or bfloat16"
    return TORCH_TO_DTYPE[dtype]


class A16Wn:
    """fp16/bf16 activations x n-bit grouped weights (reference: helper.py:187-279 ``A16Wn``)."""

    def __init__(self, device="cuda:0", dtype: Optional[torch.dtype] = None, packing_bitwidth=None,
                 post_scale=default_post_scale):
        self.device, self.dtype, self.packing_bitwidth, self.post_scale = device, dtype, packing_bitwidth, post_scale

    def from_weights(self, W_q: torch.Tensor, scales: torch.Tensor, zeros, W_nbits: int, group_size: int,
                     bias: Optional[torch.Tensor] = None) -> GemLiteLinear:
        dtype = scales.dtype if self.dtype is None else self.dtype
        gdt = _gemlite_dtype(dtype)
        out_features, in_features = W_q.shape
        W_q = W_q.to(device=self.device, dtype=torch.uint8)
        scales = scales.to(device=self.device, dtype=dtype)
        if isinstance(zeros, torch.Tensor):
            zeros = zeros.to(device=self.device, dtype=dtype)
        bias = None if bias is None else bias.to(device=self.device, dtype=dtype)
        layer = GemLiteLinear(W_nbits, group_size=group_size, in_features=in_features, out_features=out_features,
                              input_dtype=gdt, output_dtype=gdt)
        layer.pack(W_q, scales, zeros, bias=bias, fma_mode=not (self.post_scale and group_size == in_features),
                   packing_bitwidth=self.packing_bitwidth)
        return layer


class A16W8(A16Wn):
    """fp16/bf16 activations x 8-bit symmetric channel-wise weights quantised here (helper.py:88-185)."""

    def from_linear(self, linear: torch.nn.Linear) -> GemLiteLinear:
        W = linear.weight.data.to(device=self.device, dtype=torch.float32)
        dtype = linear.weight.dtype if self.dtype is None else self.dtype
        scales = (W.abs().amax(dim=1, keepdim=True) / 127.0).clamp_(min=1e-6)
        W_q = (W / scales).round_().clamp_(-128, 127).to(torch.int8)
        gdt = _gemlite_dtype(dtype)
        out_features, in_features = W.shape
        layer = GemLiteLinear(8, group_size=in_features, in_features=in_features, out_features=out_features,
                              input_dtype=gdt, output_dtype=gdt)
        bias = None if linear.bias is None else linear.bias.data.to(device=self.device, dtype=dtype)
        layer.pack(W_q, scales.to(dtype), zeros=None, bias=bias)  # unpacked int8, channel-wise -> (0, 1)
        return layer


class A8W8_dynamic:
    """8-bit dynamic activations x 8-bit channel-wise weights (reference: helper.py:405-481): int8 x int8
    or fp8 x fp8 with per-token and per-channel scales applied after the K reduction (modes (0, 3))."""

    def __init__(self, device="cuda:0", dtype: Optional[torch.dtype] = None, fp8=False, fp32_scale=True):
        self.device, self.dtype, self.fp8, self.fp32_scale = device, dtype, fp8, fp32_scale

    def from_weights(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     scales: Optional[torch.Tensor] = None) -> GemLiteLinear:
        if self.fp8:
            w_dtype = self.fp8 if isinstance(self.fp8, torch.dtype) else default_fp8
            input_dtype = TORCH_TO_DTYPE[w_dtype]
            info = torch.finfo(w_dtype)
        else:
            w_dtype, input_dtype, info = torch.int8, DType.INT8, torch.iinfo(torch.int8)
        out_features, in_features = weight.shape
        if scales is None:
            dtype = weight.dtype if self.dtype is None else self.dtype
            W = weight.to(device=self.device, dtype=torch.float32)
            scales = (W.abs().amax(dim=1, keepdim=True) / info.max).clamp_(min=1e-6)
            W_q = (W / scales).clamp_(info.min, info.max)
            W_q = W_q.to(w_dtype) if w_dtype.is_floating_point else W_q.round_().to(w_dtype)
        else:
            assert weight.element_size() == 1, "Invalid weight.dtype, should be 8-bit."
            dtype = self.dtype or (scales.dtype if scales.dtype in (torch.float16, torch.bfloat16) else torch.float16)
            W_q, scales = weight.to(self.device), scales.to(self.device)
        scales = scales.to(torch.float32 if self.fp32_scale else dtype)
        bias = None if bias is None else bias.to(device=self.device, dtype=dtype)
        layer = GemLiteLinear(8, group_size=in_features, in_features=in_features, out_features=out_features,
                              input_dtype=input_dtype, output_dtype=_gemlite_dtype(dtype), scaled_activations=True)
        layer.pack(W_q, scales, zeros=None, bias=bias)
        layer.W_group_mode, layer.channel_scale_mode = 0, 3  # post-scaling (helper.py:474-475)
        return layer

    def from_linear(self, linear: torch.nn.Linear) -> GemLiteLinear:
        return self.from_weights(linear.weight.data, None if linear.bias is None else linear.bias.data)


class A8W8_int8_dynamic(A8W8_dynamic):
    def __init__(self, device="cuda:0", dtype=None):
        super().__init__(device=device, dtype=dtype, fp8=False)


class A8W8_fp8_dynamic(A8W8_dynamic):
    def __init__(self, device="cuda:0", dtype=None):
        super().__init__(device=device, dtype=dtype, fp8=default_fp8)


A8W8_INT8_dynamic, A8W8_FP8_dynamic = A8W8_int8_dynamic, A8W8_fp8_dynamic


def patch_model(model: torch.nn.Module, processor, skip_modules=(), device="cuda:0"):
    """Replace every ``nn.Linear`` (not named in ``skip_modules``) by ``processor.from_linear(layer)``
    (reference: helper.py:34-85, Linear branch)."""
    for name, child in list(model.named_children()):
        if name in skip_modules:
            continue
        if isinstance(child, torch.nn.Linear):
            setattr(model, name, processor.from_linear(child))
        else:
            patch_model(child, processor, skip_modules, device)
    return model


def warmup(*_a, **_k):
    """The reference pre-runs Triton autotuning per shape (helper.py:1067-1118); HIP kernels are compiled
    ahead of time, so there is nothing to warm up.  (autotune_layer() below is the optional measured search.)"""
    return None


# tuning[] candidates per kernel family of libgemlite_hip (include/gemlite_hip.h: tuning[0..3]); (0,0,0,0) = planner
_TUNING_CANDIDATES = {
    "gemv": [(0, 0, 0, 0), (2, 1, 4, 0), (2, 1, 8, 0), (2, 1, 16, 0), (3, 1, 0, 0), (3, 2, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (4, 4, 0, 0)],
    "few_rows": [(0, 0, 0, 0), (1, 1, 0, 0), (2, 1, 0, 0), (2, 2, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (4, 4, 0, 0), (0, 0, 1, 0)],
    "tiled": [(0, 0, 0, 0), (0, 1, 0, 0), (0, 2, 0, 0), (0, 4, 0, 0), (0, 8, 0, 0), (0, 4, 8, 0), (0, 8, 8, 0)],
}


def autotune_layer(layer: GemLiteLinear, batch_sizes=(1,), iters: int = 50, candidates=None, verbose: bool = False) -> dict:
    """Measured search over the library's tuning knobs for one packed layer — the HIP counterpart of the reference's
    per-shape Triton autotune (core.py:559-654, helper.py:1067-1118).  For each M, every candidate that the
    library accepts is timed (device time from per-launch HIP events, `iters` launches, weights warm) and the best is
    stored in GEMLITE_HIP_CONFIG_CACHE under the reference's key; `GemLiteLinear.cache_config(path)` /
    `load_config(path)` persist and reload the table, and every later launch of that shape uses it.
    Returns {M: {"tuning": [...], "us": best, "default_us": planner}}."""
    from . import core as _core
    from ._hip import GemliteHipError
    from .bench_utils import kernel_device_us

    dev = layer.W_q.device
    in_t = _core.DTYPE_TO_TORCH[layer.input_dtype.value if not layer.scaled_activations else layer.output_dtype.value]
    meta = layer.get_meta_args()
    out = {}
    for M in batch_sizes:
        x = (torch.randn(M, layer.in_features, device=dev) / 10).to(in_t)
        scales_x = None
        if layer.scaled_activations:
            from .quant_utils import scale_activations_per_token
            x, scales_x = scale_activations_per_token(x, w_dtype=_core.DTYPE_TO_TORCH[layer.input_dtype.value])
        fam = "gemv" if M == 1 else ("few_rows" if M <= 64 else "tiled")
        best, default_us = None, None
        for cand in (candidates or _TUNING_CANDIDATES[fam]):
            try:  # device time of the kernel itself (per-launch HIP events), not host-bound wall time
                us = kernel_device_us(lambda: _core._hip_matmul(x, layer.W_q, layer.scales, layer.zeros, scales_x, meta,
                                                                -1, cand), iters=iters)
            except (NotImplementedError, GemliteHipError):
                continue  # this candidate does not apply to the shape
            if us != us:
                continue
            if cand == (0, 0, 0, 0):
                default_us = us
            if verbose:
                print(f"[autotune] M={M} tuning={cand}: {us:.2f} us")
            if best is None or us < best[1]:
                best = (cand, us)
        if best is None:
            continue
        a = _core._static_args(layer.W_q, layer.scales, layer.zeros, meta)
        key = _core.config_key(M, a.N, a.K, a.group_size, a.elements_per_sample, a.type_id)
        family = _core.config_family(-1, M, layer.W_nbits)
        entry = {"tuning": list(best[0]), "us": round(best[1], 3)}
        _core.GEMLITE_HIP_CONFIG_CACHE.setdefault(family, {})[key] = entry
        out[M] = dict(entry, default_us=None if default_us is None else round(default_us, 3))
    return out
