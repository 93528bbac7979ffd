"""Layer "processors": quantise / wrap weights into a packed GemLiteLinear (reference: gemlite/helper.py).

Processors of the reference with the same names, arguments and resulting (W_group_mode, channel_scale_mode):
A16W8 / A16W8_INT8 / A16W8_FP8, A16Wn and its ``*_HQQ_INT`` family, A8W8 dynamic (int8 / fp8), A8Wn dynamic
(fp8 activations x n-bit groups), the two BitNet processors, the MXFP / NVFP processors (A16W8/W4_MXFP, A8W8/W4_MXFP_dynamic,
A4W4_MXFP_dynamic, A4W4_NVFP_dynamic), ``patch_model``, ``cleanup_linear``.  The third-party ``hqq`` package is not part of this build: ``from_hqqlinear`` only
reads the attributes the reference reads, and the raw-tensor entry points take exactly what
``HQQLinear.unpack()`` / ``meta`` hold.
On gfx950 FP8 means OCP e4m3fn (the reference's HIP default e4m3fnuz, helper.py:13-15, is the MI300X format).
"""
from typing import Optional

import torch

from .core import GemLiteLinear
from .dtypes import TORCH_TO_DTYPE, DType
from .quant_utils import WeightQuantizerMXFP

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


class A16W8:
    """fp16/bf16 activations x 8-bit symmetric channel-wise weights, INT8 or FP8, quantised here or passed in
    pre-quantised with their scales (reference: helper.py:88-171).  `post_scale=False` folds the channel scale into
    the dequantisation (modes (2, 0)), `True` applies it after the K reduction ((0, 1))."""

    def __init__(self, device="cuda:0", dtype: Optional[torch.dtype] = None, fp8=None, fp32_scale=True, post_scale=False):
        self.device, self.dtype, self.fp8, self.fp32_scale, self.post_scale = device, dtype, fp8, fp32_scale, post_scale

    def from_weights(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     scales: Optional[torch.Tensor] = None) -> GemLiteLinear:
        weight = weight.data if isinstance(weight, torch.nn.Parameter) else weight
        bias = bias.data if isinstance(bias, torch.nn.Parameter) else bias
        out_features, in_features = weight.shape
        if scales is None:  # quantise: symmetric, one scale per output channel
            w_dtype = self.fp8 if self.fp8 else torch.int8
            info = torch.finfo(w_dtype) if w_dtype.is_floating_point else torch.iinfo(w_dtype)
            dtype = weight.dtype if self.dtype is None else self.dtype
            W = weight.to(device=self.device, dtype=torch.float32)
            scales = (W.abs().amax(dim=1, keepdim=True) / info.max).clamp_(min=1e-6)
            W_q = (W / scales).clamp_(info.min, info.max)
            W_q = W_q.to(w_dtype) if w_dtype.is_floating_point else W_q.round_().to(w_dtype)
        else:  # pre-quantised
            assert weight.element_size() == 1, f"Invalid weight.dtype, should be 8-bit (INT8 or FP8), got {weight.dtype}"
            dtype = self.dtype or (scales.dtype if scales.dtype in (torch.float16, torch.bfloat16) else torch.float16)
            W_q, scales = weight.to(self.device), scales.to(self.device)
        gdt = _gemlite_dtype(dtype)
        bias = None if bias is None else bias.to(device=self.device, dtype=dtype)
        layer = GemLiteLinear(8, group_size=in_features, in_features=in_features, out_features=out_features,
                              input_dtype=gdt, output_dtype=gdt)
        layer.pack(W_q, scales, zeros=None, bias=bias)  # scales stay as computed / given (fp32 when quantised here)
        layer.W_group_mode, layer.channel_scale_mode = (0, 1) if self.post_scale else (2, 0)
        return layer

    def from_linear(self, linear: torch.nn.Linear, del_orig: bool = True) -> GemLiteLinear:
        out = self.from_weights(linear.weight, linear.bias)
        cleanup_linear(linear, del_orig)
        return out


class A16W8_INT8(A16W8):  # (constructor arguments of the reference's aliases: helper.py:331-337)
    def __init__(self, device="cuda:0", dtype=None, fp32_scale=True, post_scale=False):
        super().__init__(device=device, dtype=dtype, fp8=None, fp32_scale=fp32_scale, post_scale=post_scale)


class A16W8_FP8(A16W8):
    def __init__(self, device="cuda:0", dtype=None, fp8=default_fp8, fp32_scale=True, post_scale=False):
        super().__init__(device=device, dtype=dtype, fp8=fp8, fp32_scale=fp32_scale, post_scale=post_scale)


class A16Wn_HQQ_INT(A16Wn):
    """A16Wn fed from an HQQ-quantised layer (reference: helper.py:339-354).  `hqq` itself is a third-party package
    that is not part of this build: `from_hqqlinear` only relies on the attributes the reference reads
    (`meta['nbits' | 'group_size' | 'shape' | 'scale' | 'zero' | 'axis']`, `unpack(dtype=)`, `bias`, `in_features`)."""

    W_nbits: Optional[int] = None

    def __init__(self, device="cuda:0", dtype=None, packing_bitwidth=None, post_scale=default_post_scale, W_nbits=None):
        super().__init__(device=device, dtype=dtype, packing_bitwidth=packing_bitwidth, post_scale=post_scale)
        if W_nbits is not None:
            self.W_nbits = W_nbits

    def from_hqqlinear(self, hqq_layer, del_orig: bool = True) -> GemLiteLinear:
        meta = hqq_layer.meta
        assert meta["axis"] == 1, "Only axis==1 is supported."
        self.device = hqq_layer.W_q.device
        group_size = meta["group_size"] or hqq_layer.in_features
        W_q = hqq_layer.unpack(dtype=torch.uint8).view(meta["shape"])
        scales, zeros = meta["scale"].clone(), meta["zero"].clone()
        bias = None if hqq_layer.bias is None else hqq_layer.bias.clone()
        cleanup_linear(hqq_layer, del_orig)
        return self.from_weights(W_q, scales, zeros, meta["nbits"], group_size, bias=bias)


class A16W8_HQQ_INT(A16Wn_HQQ_INT):
    W_nbits = 8


class A16W4_HQQ_INT(A16Wn_HQQ_INT):
    W_nbits = 4


class A16W2_HQQ_INT(A16Wn_HQQ_INT):
    W_nbits = 2


class A16W1_HQQ_INT(A16Wn_HQQ_INT):
    W_nbits = 1


class A8Wn_HQQ_INT_dynamic(A16Wn_HQQ_INT):
    """8-bit (FP8 by default) dynamically quantised activations x n-bit grouped weights: (q - z) * s dequant with the
    per-token scale applied after the K reduction — modes (3, 2); channel-wise weights move their scale to the
    epilogue as well ((1, 3) with `post_scale`, else (3, 2)).  Reference: helper.py:502-615."""

    def __init__(self, device="cuda:0", packing_bitwidth=None, dtype=None, post_scale=default_post_scale, fp8=default_fp8,
                 fp32_scale=False, W_nbits=None):
        assert W_nbits is not None or self.W_nbits is not None, "W_nbits should be given (8, 4, 2 ...)"
        super().__init__(device=device, dtype=dtype, packing_bitwidth=packing_bitwidth, post_scale=post_scale, W_nbits=W_nbits)
        self.fp8, self.fp32_scale = fp8, fp32_scale

    def from_weights(self, W_q, scales, zeros, W_nbits=None, group_size=None, bias=None) -> GemLiteLinear:
        W_q = W_q.data if isinstance(W_q, torch.nn.Parameter) else W_q
        W_nbits = self.W_nbits if W_nbits is None else W_nbits
        group_size = W_q.numel() // scales.numel() if group_size is None else group_size
        dtype = self.dtype or (scales.dtype if scales.dtype in (torch.float16, torch.bfloat16) else torch.float16)
        out_features, in_features = W_q.shape
        layer = GemLiteLinear(W_nbits, group_size=group_size, in_features=in_features, out_features=out_features,
                              input_dtype=TORCH_TO_DTYPE[self.fp8], output_dtype=_gemlite_dtype(dtype), scaled_activations=True)
        layer.pack(W_q.to(device=self.device, dtype=torch.uint8),
                   scales.to(device=self.device, dtype=torch.float32 if self.fp32_scale else dtype),
                   None if zeros is None else zeros.to(device=self.device, dtype=dtype),
                   bias=None if bias is None else bias.to(device=self.device, dtype=dtype),
                   packing_bitwidth=self.packing_bitwidth, fma_mode=False)
        if group_size == in_features:
            layer.W_group_mode, layer.channel_scale_mode = (1, 3) if self.post_scale else (3, 2)
        return layer


class A8W4_HQQ_INT_dynamic(A8Wn_HQQ_INT_dynamic):
    W_nbits = 4


class A8W2_HQQ_INT_dynamic(A8Wn_HQQ_INT_dynamic):
    W_nbits = 2


class A16W158_INT:
    """BitNet b1.58: ternary weights {-1, 0, 1} stored as 2-bit codes {0, 1, 2} with a scalar zero of 1 and one scale
    for the whole matrix applied per output channel after the K reduction — modes (1, 1).  Reference: helper.py:950-1004."""

    scaled_activations, channel_scale_mode = False, 1

    def __init__(self, device="cuda:0", dtype: Optional[torch.dtype] = None, fp32_scale=True):
        self.device, self.dtype, self.fp32_scale = device, dtype, fp32_scale

    def from_weights(self, weight: torch.Tensor, weight_scale, bias: Optional[torch.Tensor] = None) -> GemLiteLinear:
        weight = weight.data if isinstance(weight, torch.nn.Parameter) else weight
        bias = bias.data if isinstance(bias, torch.nn.Parameter) else bias
        dtype = weight.dtype if self.dtype is None else self.dtype
        W_q = (weight.to(device=self.device, dtype=dtype) + 1).to(torch.uint8)
        out_features, in_features = W_q.shape
        ws = float(weight_scale.item() if isinstance(weight_scale, torch.Tensor) else weight_scale)
        scales = torch.full((out_features, 1), ws, dtype=torch.float32 if self.fp32_scale else dtype, device=self.device)
        layer = GemLiteLinear(2, group_size=in_features, in_features=in_features, out_features=out_features,
                              input_dtype=DType.INT8 if self.scaled_activations else _gemlite_dtype(dtype),
                              output_dtype=_gemlite_dtype(dtype), scaled_activations=self.scaled_activations)
        layer.pack(W_q, scales=scales, zeros=1, bias=None if bias is None else bias.to(device=self.device, dtype=dtype))
        layer.W_group_mode, layer.channel_scale_mode = 1, self.channel_scale_mode  # shift only + post-scale
        return layer

    def from_bitlinear(self, linear_layer, del_orig: bool = True) -> GemLiteLinear:
        out = self.from_weights(linear_layer.weight, linear_layer.weight_scale, linear_layer.bias)
        cleanup_linear(linear_layer, del_orig)
        return out


class A8W158_INT_dynamic(A16W158_INT):
    """BitNet with int8 dynamically quantised activations: modes (1, 3).  Reference: helper.py:1006-1062."""

    scaled_activations, channel_scale_mode = True, 3


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
    def __init__(self, device="cuda:0", dtype=None, fp8=default_fp8):  # reference: helper.py:491-496 (its AMD default is the
        super().__init__(device=device, dtype=dtype, fp8=fp8)          # MI300X fnuz format; gfx950 MFMA is OCP e4m3fn)


A8W8_INT8_dynamic, A8W8_FP8_dynamic = A8W8_int8_dynamic, A8W8_fp8_dynamic


# ------------------------------------------------------------------------------------------------------
# block-scaled formats (reference: helper.py:173-331 quant_type="MXFP", :372-400, :658-950)
# ------------------------------------------------------------------------------------------------------
class _BlockScaledProcessor:
    """Shared body of the MXFP / NVFP processors.  ``from_weights(weight, bias, scales)`` takes pre-quantised
    elements (fp8 tensor, or uint8 e2m1 codes ``[N, K]``) with their block scales ``[N, K / group]``;
    ``from_linear`` quantises an ``nn.Linear`` with ``WeightQuantizerMXFP`` first.  Subclasses fix: W_nbits, group size,
    the layer's input format, whether activations are quantised, and the channel_scale_mode set after ``pack()``."""
    W_nbits = None
    group_size = 32
    scaled_activations = True

    def __init__(self, device="cuda:0", dtype: Optional[torch.dtype] = None):
        self.device, self.dtype = device, dtype
        self.quantizer_mx = None
        self.mx_fp8_dtype = default_fp8

    # -- per-format hooks ------------------------------------------------------------------------------
    def _input_dtype(self, dtype: torch.dtype) -> DType:
        raise NotImplementedError

    def _channel_scale_mode(self) -> int:
        return 4

    def _quantize(self, W: torch.Tensor):
        if self.W_nbits == 8:
            return self.quantizer_mx.quantize_mxfp8(W, index=True, mx_fp8_dtype=self.mx_fp8_dtype)
        return self.quantizer_mx.quantize_mxfp4(W, index=True)

    # -- API of the reference processors ------------------------------------------------------------
    def from_weights(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     scales: Optional[torch.Tensor] = None) -> GemLiteLinear:
        if isinstance(weight, torch.nn.Parameter):
            weight = weight.data
        if isinstance(bias, torch.nn.Parameter):
            bias = bias.data
        assert scales is not None, "Scales parameter cannot be None. Use from_linear() call to pre-quantize the weights."
        if self.W_nbits == 8:
            assert weight.is_floating_point() and weight.element_size() == 1, f"Invalid weight.dtype, should be an MXFP8 dtype, got {weight.dtype}."
        else:
            assert weight.dtype == torch.uint8, f"Invalid weight.dtype, should be uint8 (e2m1 codes), got {weight.dtype}."
        if self.group_size == 16:
            assert scales.dtype == torch.float8_e4m3fn, f"Invalid scales.dtype, should be float8_e4m3fn, got {scales.dtype}."
        else:
            assert scales.dtype in (torch.float8_e8m0fnu, torch.uint8), f"Invalid scales.dtype, should be e8m0 / view(uint8), got {scales.dtype}."
        dtype = self.dtype
        if dtype is None:
            assert not self.scaled_activations, "Input dtype should be either torch.float16 or torch.bfloat16, not None."
            dtype = torch.float16
        out_features, in_features = weight.shape
        W_q = weight.to(device=self.device)
        scales = scales.to(device=self.device).view(out_features, in_features // self.group_size)
        bias = None if bias is None else bias.to(device=self.device, dtype=dtype)
        layer = GemLiteLinear(self.W_nbits, group_size=self.group_size, in_features=in_features, out_features=out_features,
                              input_dtype=self._input_dtype(dtype),
                              # weight-only layers carry the MX code in both slots (helper.py:252-253)
                              output_dtype=_gemlite_dtype(dtype) if self.scaled_activations else self._input_dtype(dtype),
                              scaled_activations=self.scaled_activations)
        layer.pack(W_q, scales, zeros=None, bias=bias)
        if self.scaled_activations:  # helper.py:703-705, 779-781, 856-857, 922-923
            layer.W_group_mode, layer.channel_scale_mode = 0, self._channel_scale_mode()
        return layer

    def from_linear(self, linear_layer: torch.nn.Linear, del_orig: bool = True) -> GemLiteLinear:
        if self.quantizer_mx is None:
            self.quantizer_mx = WeightQuantizerMXFP(device=self.device, compute_dtype=linear_layer.weight.dtype)
        W = linear_layer.weight.data
        bias = None if linear_layer.bias is None else linear_layer.bias.clone()
        N, K = W.shape
        W_q, scales = self._quantize(W)
        W_q, scales = W_q.view(N, K), scales.view(N, K // self.group_size)
        cleanup_linear(linear_layer, del_orig)
        return _BlockScaledProcessor.from_weights(self, weight=W_q, bias=bias, scales=scales)


class A16Wn_MXFP(_BlockScaledProcessor):
    """fp16 / bf16 activations x MXFP8 / MXFP4 weights (reference: helper.py:372-400; layer dtype MXFP16 / MXBF16)."""
    scaled_activations = False

    def __init__(self, device="cuda:0", dtype=None, W_nbits=None):
        super().__init__(device=device, dtype=dtype)
        self.W_nbits = W_nbits

    def _input_dtype(self, dtype):
        if dtype == torch.float16:
            return DType.MXFP16
        if dtype == torch.bfloat16:
            return DType.MXBF16
        raise Exception(f"Unsupported dtype for MXFP. Got {dtype}, supported [torch.float16, torch.bfloat16]")

    def from_weights(self, W_q, scales, bias=None):  # the reference's argument order for this family
        return super().from_weights(weight=W_q, bias=bias, scales=scales)


class A16W8_MXFP(A16Wn_MXFP):
    def __init__(self, device="cuda:0", dtype=None):
        super().__init__(device=device, dtype=dtype, W_nbits=8)


class A16W4_MXFP(A16Wn_MXFP):
    def __init__(self, device="cuda:0", dtype=None):
        super().__init__(device=device, dtype=dtype, W_nbits=4)


class A8Wn_MXFP_dynamic(_BlockScaledProcessor):
    """MXFP8 activations (quantised per call) x MXFP8 / MXFP4 weights (reference: helper.py:732-815).  ``post_scale=True``:
    one fp32 scale per token applied after the K reduction (channel_scale_mode 2); ``False``: e8m0 microscales per 32 k
    inside the contraction (4)."""

    def __init__(self, device="cuda:0", dtype=None, post_scale=True, fp8=default_fp8, W_nbits=None):
        assert W_nbits is not None, "W_nbits argument should be either 8 or 4, not None."
        super().__init__(device=device, dtype=dtype)
        self.mx_fp8_dtype, self.post_scale, self.W_nbits = fp8, post_scale, W_nbits

    def _input_dtype(self, dtype):
        return DType.MXFP8

    def _channel_scale_mode(self):
        return 2 if self.post_scale else 4


class A8W8_MXFP_dynamic(A8Wn_MXFP_dynamic):
    def __init__(self, device="cuda:0", dtype=None, post_scale=True, fp8=default_fp8):
        super().__init__(device=device, dtype=dtype, post_scale=post_scale, fp8=fp8, W_nbits=8)


class A8W4_MXFP_dynamic(A8Wn_MXFP_dynamic):
    def __init__(self, device="cuda:0", dtype=None, post_scale=True, fp8=default_fp8):
        super().__init__(device=device, dtype=dtype, post_scale=post_scale, fp8=fp8, W_nbits=4)


class A4W4_MXFP_dynamic(_BlockScaledProcessor):
    """MXFP4 activations x MXFP4 weights (reference: helper.py:816-880)."""
    W_nbits = 4

    def __init__(self, device="cuda:0", dtype=None):
        super().__init__(device=device, dtype=dtype)
        self.input_dtype = DType.MXFP4

    def _input_dtype(self, dtype):
        return self.input_dtype


class A4W4_NVFP_dynamic(_BlockScaledProcessor):
    """NVFP4 activations x NVFP4 weights: e2m1 elements, e4m3 scale per 16 k, meta scale 0.05 (reference:
    helper.py:882-946).  gfx950's matrix core has no instruction for this format: the layer runs on the coverage kernel."""
    W_nbits = 4
    group_size = 16

    def __init__(self, device="cuda:0", dtype=None):
        super().__init__(device=device, dtype=dtype)
        self.input_dtype = DType.NVFP4

    def _input_dtype(self, dtype):
        return self.input_dtype

    def _quantize(self, W):
        return self.quantizer_mx.quantize_nvfp4(W, index=True)


def cleanup_linear(linear_layer, del_orig: bool = True):
    """Drop the original tensors of a layer that has been converted (reference: helper.py:25-31)."""
    if del_orig:
        for attr in ("weight", "bias", "weight_scale", "W_q", "meta"):
            val = getattr(linear_layer, attr, None)
            if val is not None and hasattr(val, "__len__") and len(val) > 0:
                setattr(linear_layer, attr, None)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def patch_model(model: torch.nn.Module, device, processor=None, skip_modules=("lm_head", "vision", "visual"), group_size=64):
    """Replace every ``nn.Linear`` whose qualified name contains none of ``skip_modules`` by the processor's layer
    (reference: helper.py:34-85; same argument order).  HQQ processors (``from_hqqlinear``) would first quantise
    with the third-party ``hqq`` package, which this build does not ship: they raise NotImplementedError here — feed
    ``from_weights`` / ``from_hqqlinear`` with already quantised tensors instead."""
    if processor is None or hasattr(device, "from_linear") or hasattr(device, "from_hqqlinear"):
        device, processor = (processor if processor is not None else "cuda:0"), device  # (model, processor[, device]) order
    if not hasattr(processor, "from_linear"):
        raise NotImplementedError("this processor needs layers quantised by the `hqq` package (not part of this build)")

    def _walk(module, prefix):
        for name, child in list(module.named_children()):
            full = f"{prefix}.{name}" if prefix else name
            if isinstance(child, torch.nn.Linear):
                if not any(sk in full for sk in skip_modules):
                    if hasattr(processor, "device"):
                        processor.device = device
                    setattr(module, name, processor.from_linear(child.to(device)))
            else:
                _walk(child, full)

    _walk(model, "")
    return model.to(device)


def warmup(processor=None, shapes=(), batch_sizes=None, group_size: int = 64, dtype: torch.dtype = torch.float16):
    """The reference pre-runs Triton autotuning per (shape, M bucket) here (helper.py:1067-1118).  The HIP kernels are compiled ahead of
    time, so what is left to warm is the first-use work of a serving process: loading the code object, the shipped tuning table, and —
    with a processor and shapes, same call as the reference's `warmup(A8W8_INT8_dynamic(), shapes=[(4096, 4096)], batch_sizes=[1, 8])` —
    one call per (out_features, in_features) and batch size on a random layer of that shape, which sizes the stream's workspace for the
    largest plan, raises the kernels' LDS limits and fills the launch-template caches, so that no request pays for them.
    (`group_size` belongs to the reference's hqq path; processors that need hqq are skipped.  autotune_layer() below is the measured
    search over the planners' alternatives.)"""
    from . import _hip
    from .core import _M_BUCKETS, autoload_default_config
    import logging
    logger = logging.getLogger(__name__)
    _hip.load()
    if not torch.cuda.is_available():
        return None
    dev = torch.device("cuda", torch.cuda.current_device())
    autoload_default_config(dev.index)
    if processor is None or not shapes or not hasattr(processor, "from_linear"):
        return None
    if batch_sizes is None:
        batch_sizes = _M_BUCKETS[::-1]  # the reference's default: every M bucket (helper.py:1067)
    for out_features, in_features in shapes:
        linear = torch.nn.Linear(in_features, out_features, bias=False, device=dev, dtype=dtype)
        try:
            layer = processor.from_linear(linear)
        except Exception as e:  # (a processor that only converts HQQ layers, a shape it rejects: same as the reference — say so, go on)
            logger.warning(f"warmup: {type(processor).__name__} does not take a {out_features} x {in_features} layer: {e}")
            continue
        for bs in batch_sizes:
            layer(torch.randn(int(bs), in_features, device=dev, dtype=dtype) / 10)
        del layer, linear
    torch.cuda.synchronize()
    return None


# tuning[] candidates per kernel family of libgemlite_hip (include/gemlite_hip.h: tuning[0..3]); (0,0,0,0) = planner
_TUNING_CANDIDATES = {
    "gemv": [(0, 0, 0, 0), (2, 1, 4, 0), (2, 1, 8, 0), (2, 1, 16, 0), (3, 1, 0, 0), (3, 2, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (4, 4, 0, 0)],
    # (round 4: 4 = the 16-column rows kernels of the 8-bit families, 2 = their tile kernels, 5 = unsplit 64 x 64 A8W8 tiles, 7 = the
    #  streaming kernels of rounds 1-3; candidates a family does not know fall through to its planner or are refused, never mis-run)
    "few_rows": [(0, 0, 0, 0), (1, 1, 0, 0), (2, 1, 0, 0), (2, 2, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (4, 4, 0, 0), (0, 0, 1, 0),
                 (3, 0, 0, 0), (3, 0, 1, 0), (3, 0, 2, 0), (4, 0, 0, 0), (2, 0, 0, 0), (5, 0, 0, 0), (7, 0, 0, 0)],
    # 8-wave MFMA kernel: tuning[1] = K slices, tuning[2] = tile rows / 32 (32 / 34: the narrow 64-column tiles of round 4)
    "tiled": [(0, 0, 0, 0)] + [(0, sk, mi, 0) for mi in (1, 2, 4, 8) for sk in (1, 2, 3, 4, 6, 8)] +
             [(0, 1, 32, 0), (0, 2, 32, 0), (0, 1, 34, 0), (0, 2, 34, 0), (4, 0, 0, 0), (5, 0, 0, 0), (5, 0, 2, 0), (6, 0, 0, 0)],
    # block-scaled layers, 2 <= M (include/gemlite_hip.h): [0] 2 = 8-wave tile kernels, 3 = 256 x 256 tiles, 4 = few-row kernel,
    # 6 = unsplit 64 x 64 tiles ([2] = stages); [1] K slices, [2] tile rows / 32 or 32 = narrow tiles (16-bit activations, NVFP4)
    "mx": [(0, 0, 0, 0), (2, 0, 0, 0), (3, 0, 0, 0), (4, 0, 0, 0), (6, 0, 0, 0), (6, 0, 2, 0), (6, 0, 4, 0), (0, 1, 32, 0), (0, 2, 32, 0)] +
          [(0, sk, mi, 0) for mi in (1, 2, 4) for sk in (1, 2, 4)],
}


def autotune_layer(layer: GemLiteLinear, batch_sizes=(1,), iters: int = 50, candidates=None, verbose: bool = False,
                   cold: bool = False) -> dict:
    """Measured search over the library's tuning knobs for one packed layer — the HIP counterpart of the reference's
    per-shape Triton autotune (core.py:559-654, helper.py:1067-1118).  For each M, every candidate that the
    library accepts is timed (device time from per-launch HIP events, `iters` launches, weights warm) and the best is
    stored in GEMLITE_HIP_CONFIG_CACHE under the reference's key (`cold=True` rewrites a 512 MiB buffer before every timed
    launch so that the weights come from HBM, as they do inside a model, instead of the 256 MiB Infinity Cache); `GemLiteLinear.cache_config(path)` /
    `load_config(path)` persist and reload the table, and every later launch of that shape uses it.
    Returns {M: {"tuning": [...], "us": best, "default_us": planner, "candidates": {tuning: us}}}."""
    from . import core as _core
    from ._hip import GemliteHipError
    from .bench_utils import kernel_device_us

    mx = bool(_core.is_mx_dtype(layer.input_dtype.value))
    dev = layer.W_q.device
    in_t = _core.DTYPE_TO_TORCH[layer.output_dtype.value if (mx or layer.scaled_activations) else layer.input_dtype.value]
    meta = layer.get_meta_args()
    out = {}
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if cold else None
    for M in batch_sizes:
        x = (torch.randn(M, layer.in_features, device=dev) / 10).to(in_t)
        scales_x = None
        if mx:
            # block-scaled layers (round 4): entries are filed under the `mx` families lookup_tuning() reads.  One row of a dynamic layer is
            # quantised INSIDE the few-row kernel, which a table entry would switch off: not tuned
            code, c_mode = layer.input_dtype.value, layer.channel_scale_mode
            if bool(meta[0]):
                if M == 1:
                    continue
                from . import quant_utils as _q
                if code == _core.DType.MXFP8.value and c_mode == 4:
                    x, scales_x = _q.scale_activations_mxfp8(x, w_dtype=torch.float8_e4m3fn)
                elif code == _core.DType.MXFP8.value and c_mode == 2:
                    x, scales_x = _q.scale_activations_per_token(x, w_dtype=torch.float8_e4m3fn)
                elif code == _core.DType.MXFP4.value and c_mode == 4:
                    x, scales_x = _q.scale_activations_mxfp4(x)
                elif code == _core.DType.NVFP4.value and c_mode == 4:
                    x, scales_x = _q.scale_activations_nvfp4(x)
                else:
                    raise NotImplementedError(f"autotune_layer: no activation quantiser for {layer.input_dtype} with channel_scale_mode {c_mode}")
        elif layer.scaled_activations:
            from .quant_utils import scale_activations_per_token
            x, scales_x = scale_activations_per_token(x, w_dtype=_core.DTYPE_TO_TORCH[layer.input_dtype.value])
        fam = "mx" if mx else ("gemv" if M == 1 else ("few_rows" if M <= 32 else "tiled"))
        best, default_us, timed = None, None, {}
        cands = list(candidates or _TUNING_CANDIDATES[fam])
        if candidates is None and not mx and M >= 2:
            # round 5: the decode-shaped rows kernel of 4-bit words under 16-bit activations ([0] = 9; [1] = 1 / 2 column tiles per block)
            # and the unsplit 128 x 128 tiles of unpacked 8-bit layers ([0] = 10; [2] = stages)
            if layer.W_nbits == 4 and layer.elements_per_sample == 8 and not layer.scaled_activations:
                cands += [(9, 0, 0, 0), (9, 1, 0, 0), (9, 2, 0, 0)]
            if layer.W_nbits == 2 and layer.elements_per_sample == 16 and not layer.scaled_activations:
                cands += [(9, 0, 0, 0)]
            if layer.elements_per_sample == 1 and layer.scaled_activations and M > 64:
                cands += [(10, 0, 0, 0), (10, 0, 5, 0)]
        for cand in cands:
            try:  # device time of the kernel itself (per-launch HIP events), not host-bound wall time
                us = kernel_device_us(lambda: _core._hip_matmul(x, layer.W_q, layer.scales, layer.zeros, scales_x, meta,
                                                                -1, cand), iters=iters,
                                      before_each=(lambda: flush.fill_(1)) if flush is not None else None)
            except (NotImplementedError, GemliteHipError):
                continue  # this candidate does not apply to the shape
            if us != us:
                continue
            timed[str(tuple(cand))] = round(us, 3)
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
        family = _core.config_family(-1, M, layer.W_nbits, mx)
        entry = {"tuning": list(best[0]), "us": round(best[1], 3)}
        _core.GEMLITE_HIP_CONFIG_CACHE.setdefault(family, {})[key] = entry
        out[M] = dict(entry, default_us=None if default_us is None else round(default_us, 3), candidates=timed)
    return out
