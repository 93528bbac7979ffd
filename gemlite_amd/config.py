"""Tuning / caching switches with the reference's names (gemlite/triton_kernels/config.py:9-57).

There is no Triton autotuner behind them: the HIP library chooses the kernel variant per shape in C
(`gemlite_hip_forward`, tuning[] hints).  The setters keep the reference's call signatures so user code
that configures GemLite at start-up keeps working; what they change here is documented per setter.
"""
from typing import Union

MATMUL_DTYPES = ["GEMV", "GEMV_REVSPLITK", "GEMV_SPLITK", "GEMM_SPLITK", "GEMM"]


class AUTOTUNE:
    GEMV = "fast"
    GEMV_REVSPLITK = "fast"
    GEMV_SPLITK = "fast"
    GEMM_SPLITK = "fast"
    GEMM = "fast"
    USE_CUDA_GRAPH = False


class KERNEL:
    # The reference's output-ring cache (gemv_revsplitK_kernels.py:405-419) exists to skip a per-call
    # torch.zeros(); the HIP GEMV needs no zero-initialised output, so the flag is accepted and unused.
    ENABLE_CACHING = False
    CACHE_SIZE = 256


def set_kernel_caching(enable: bool):
    KERNEL.ENABLE_CACHING = bool(enable)


def set_autotune(config: Union[dict, str, bool], **kwargs):
    """Record the requested mode per kernel family ("max" / "fast" / "default").  Kernel variants are
    selected inside libgemlite_hip; nothing is recompiled or reloaded."""
    if isinstance(config, str):
        for key in MATMUL_DTYPES:
            setattr(AUTOTUNE, key, config.lower())
    elif isinstance(config, bool):
        for key in MATMUL_DTYPES:
            setattr(AUTOTUNE, key, "max" if config else "default")
    elif isinstance(config, dict):
        for key, val in config.items():
            setattr(AUTOTUNE, key, val)
    if "use_cuda_graph" in kwargs:
        AUTOTUNE.USE_CUDA_GRAPH = bool(kwargs["use_cuda_graph"])
