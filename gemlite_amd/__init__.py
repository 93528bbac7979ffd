"""gemlite_amd — MI355X-native (gfx950 / CDNA4) drop-in for the GemLite low-bit matmul hot path.

Same public names as ``gemlite/__init__.py:5-22`` of the reference; compute goes through the hand-written
HIP kernels in ``gemlite_amd/csrc`` behind the C ABI of ``include/gemlite_hip.h``.
Usage: ``import gemlite_amd as gemlite``.
"""
__version__ = "0.1.0"

from .core import (  # noqa: F401
    GEMLITE_ACC_DTYPE,
    DType,
    GemLiteLinear,
    GemLiteLinearHIP,
    GemLiteLinearTriton,
    forward_functional,
    get_matmul_type,
    set_acc_dtype,
    set_autotune,
    set_autotune_setting,
    set_kernel_caching,
    set_packing_bitwidth,
)
from . import helper  # noqa: F401,E402

load_config = GemLiteLinear.load_config
cache_config = GemLiteLinear.cache_config
reset_config = GemLiteLinear.reset_config
