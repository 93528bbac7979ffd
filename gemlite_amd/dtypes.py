"""Integer dtype codes that travel in ``meta_args`` / ``state_dict['metadata']``.

Wire-format parity with the reference enum (reference: gemlite/dtypes.py:8-29 for the
codes, :32-71 for the torch maps, :95-111 for packing dtypes and dtype families).  The
codes are part of the drop-in contract (they are serialized), so the numbers must match;
everything else here (how the tables are built, the helpers) is this package's own.
No Triton dependency: the HIP backend receives the raw integer codes through the C ABI
(include/gemlite_hip.h, ``gemlite_dtype_t``).
"""
from enum import Enum

import torch

# (name, code, torch dtype, is the canonical reverse mapping for that torch dtype)
_SPEC = (
    ("FP32", 0, torch.float32, True),
    ("FP16", 1, torch.float16, True),
    ("BF16", 2, torch.bfloat16, True),
    ("FP8", 3, torch.float8_e4m3fn, True),
    ("INT8", 4, torch.int8, True),
    ("UINT8", 5, torch.uint8, True),
    ("INT32", 6, torch.int32, True),
    ("UINT32", 7, torch.uint32, True),
    ("FP8e5", 8, torch.float8_e5m2, True),
    ("INT16", 9, torch.int16, True),
    ("UINT16", 10, torch.uint16, True),
    ("INT64", 11, torch.int64, True),
    ("FP8e4nuz", 12, torch.float8_e4m3fnuz, True),
    ("FP8e5nuz", 13, torch.float8_e5m2fnuz, True),
    ("MXFP16", 14, torch.float16, False),
    ("MXBF16", 15, torch.bfloat16, False),
    ("MXFP8", 16, torch.float8_e4m3fn, False),
    ("MXFP4", 17, torch.uint8, False),
    ("NVFP4", 18, torch.uint8, False),
    ("E8M0", 19, torch.float8_e8m0fnu, True),
)

_members = {name: code for name, code, _, _ in _SPEC}
_members["FP8e4"] = _members["FP8"]  # alias, same code (reference dtypes.py:12-13)
DType = Enum("DType", _members)
DType.__doc__ = "dtype codes shared with the C ABI (gemlite_dtype_t)"

DTYPE_TO_TORCH = {code: tdt for _, code, tdt, _ in _SPEC}
TORCH_TO_DTYPE = {tdt: DType(code) for _, code, tdt, canon in _SPEC if canon}

PACKING_BITWIDTH_TO_TORCH_DTYPE = {
    8: torch.uint8,
    16: torch.int16,
    32: torch.int32,
    64: torch.int64,
}

FP8_DTYPES = [DType.FP8, DType.FP8e4, DType.FP8e5, DType.FP8e4nuz, DType.FP8e5nuz]
FP8_INT8_DTYPES = [DType.INT8] + FP8_DTYPES
MX_DTYPES = [DType.MXFP16, DType.MXBF16, DType.MXFP8, DType.MXFP4, DType.NVFP4]
MX_DTYPES_val = [d.value for d in MX_DTYPES]


def is_mx_dtype(input_dtype):
    """True for the microscaling codes; accepts the enum or its integer (dtypes.py:107-111)."""
    if isinstance(input_dtype, DType):
        return input_dtype in MX_DTYPES
    if isinstance(input_dtype, int):
        return input_dtype in MX_DTYPES_val
    return None


def dtype_itemsize(code: int) -> int:
    return torch.empty((), dtype=DTYPE_TO_TORCH[int(code)]).element_size()
