"""`gemlite.triton_kernels.utils` — the host-side helpers of the reference's module (utils.py:99-183) that callers use; the Triton
device functions of that file (dequantize, swizzle_tile_*, atomic_add_cas) have no counterpart: they live inside the HIP kernels."""
import torch

from ..core import _M_BUCKETS, get_closest_m  # noqa: F401  (utils.py:136-174: the autotune M buckets)

M_MAXVAL = 4096  # utils.py:171
# utils.py:172: M -> bucket, for every M up to M_MAXVAL (the reference precomputes the same table)
M_MAPPING = {m: get_closest_m(m) for m in range(M_MAXVAL + 1)}

IS_HIP = True  # utils.py:108-109, 182: torch.version.hip is set — this package only runs on ROCm
NATIVE_ATOMIC = False  # utils.py:125-128, 183: hard-wired False in the reference (bf16 atomic add)


def is_hip():
    return True


def next_power_of_2(v):
    """utils.py:102-103."""
    return 1 if v < 1 else 1 << (int(v) - 1).bit_length()


def is_divisible(dividend, divisor):
    """utils.py:105-106."""
    return dividend % divisor == 0


def get_num_SMs(device):
    """utils.py:131-134: compute units of `device`."""
    return torch.cuda.get_device_properties(device).multi_processor_count


def gpu_supports_bfloat16_atomicadd():
    return False


def gpu_has_more_shared_memory(ref_gpus=("a100", "h100", "h200", "h20", "h800", "b100", "b200")):
    """utils.py:111-113 matches NVIDIA names; an MI355X CU has 160 KiB of LDS — more than any of them."""
    return True


def gpu_supports_float16_acc(*_a, **_k):
    """utils.py:115-122: fp16 accumulation only on a list of GeForce parts; never on AMD."""
    return False


def get_gpu_shared_memory():
    """utils.py:176-180: bytes of shared memory per block; gfx950: 160 KiB of LDS."""
    return 163840
