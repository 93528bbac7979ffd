"""`gemlite.triton_kernels.config` (reference: gemlite/triton_kernels/config.py:9-57): the very objects of gemlite_amd.config, so a
flag flipped through either import path is seen by both."""
from ..config import AUTOTUNE, KERNEL, MATMUL_DTYPES, set_autotune, set_kernel_caching  # noqa: F401


def reload_all_modules():
    """The reference re-imports its Triton kernel modules so that new autotune settings take effect (config.py:25-39); the HIP
    library picks its kernel per launch, there is nothing to reload."""
    return None
