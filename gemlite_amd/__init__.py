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
from . import triton_kernels  # noqa: F401,E402  (import-path shims: gemlite.triton_kernels.config / .utils)


def alias_as_gemlite(force: bool = False):
    """Make `import gemlite`, `from gemlite.core import ...`, `from gemlite.triton_kernels.config import KERNEL` resolve to this
    package (a drop-in switch for code written against the reference; its own test files import exactly these paths).  Refuses to
    shadow an already imported real `gemlite` unless `force`."""
    import sys
    from . import bitpack, config, core, dtypes, helper, quant_utils
    if "gemlite" in sys.modules and sys.modules["gemlite"] is not sys.modules[__name__] and not force:
        raise RuntimeError("a different `gemlite` is already imported; pass force=True to replace it")
    me = sys.modules[__name__]
    sys.modules["gemlite"] = me
    for sub, mod in (("core", core), ("helper", helper), ("dtypes", dtypes), ("bitpack", bitpack), ("quant_utils", quant_utils),
                     ("config", config), ("triton_kernels", triton_kernels), ("triton_kernels.config", triton_kernels.config),
                     ("triton_kernels.utils", triton_kernels.utils)):
        sys.modules["gemlite." + sub] = mod
    return me
