"""ctypes binding of libgemlite_hip.so (C ABI: include/gemlite_hip.h).

This is the ONLY compute backend of the package: there is no eager-PyTorch or CPU fallback.  If the
shared library is missing or a launch is refused, a Python exception is raised (loudly) — the same
exception classes the reference raises for the same conditions (SURVEY.md §8(b) "Errors").
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgemlite_hip.so")
ABI_VERSION = 1

# status codes (gemlite_status_t)
OK, ERR_BAD_ARGUMENT, ERR_UNSUPPORTED, ERR_BAD_SHAPE, ERR_WORKSPACE, ERR_LAUNCH, ERR_NO_DEVICE, ERR_NO_FUSED_QUANT = 0, -1, -2, -3, -4, -5, -6, -7


class ForwardArgs(C.Structure):
    """struct gemlite_hip_forward_args (field order/types must match the header)."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("matmul_type", C.c_int32),
        ("x", C.c_void_p),
        ("w_q", C.c_void_p),
        ("scales", C.c_void_p),
        ("zeros", C.c_void_p),
        ("scales_x", C.c_void_p),
        ("out", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_uint64),
        ("M", C.c_int64),
        ("N", C.c_int64),
        ("K", C.c_int64),
        ("W_nbits", C.c_int32),
        ("group_size", C.c_int32),
        ("unpack_mask", C.c_int32),
        ("elements_per_sample", C.c_int32),
        ("w_pack_bits", C.c_int32),
        ("w_dtype", C.c_int32),
        ("input_dtype", C.c_int32),
        ("output_dtype", C.c_int32),
        ("acc_dtype", C.c_int32),
        ("meta_dtype", C.c_int32),
        ("zeros_dtype", C.c_int32),
        ("channel_scale_mode", C.c_int32),
        ("W_group_mode", C.c_int32),
        ("zero_is_scalar", C.c_int32),
        ("data_contiguous", C.c_int32),
        ("type_id", C.c_int32),
        ("stride_xm", C.c_int64),
        ("stride_xk", C.c_int64),
        ("stride_wk", C.c_int64),
        ("stride_wn", C.c_int64),
        ("stride_om", C.c_int64),
        ("stride_on", C.c_int64),
        ("stride_meta_g", C.c_int64),
        ("stride_meta_n", C.c_int64),
        ("stride_sx_m", C.c_int64),
        ("tuning", C.c_int32 * 4),
    ]


_lib = None
_lib_lock = threading.Lock()


class GemliteHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GemliteHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gemlite_amd/csrc`.  gemlite_amd has no non-HIP fallback."
            )
        lib = C.CDLL(LIB_PATH)
        lib.gemlite_hip_abi_version.restype = C.c_int
        lib.gemlite_hip_build_info.restype = C.c_char_p
        lib.gemlite_hip_status_string.restype = C.c_char_p
        lib.gemlite_hip_status_string.argtypes = [C.c_int]
        lib.gemlite_hip_last_hip_error.restype = C.c_int
        for name in ("gemlite_hip_query", "gemlite_hip_forward"):
            getattr(lib, name).restype = C.c_int
        lib.gemlite_hip_query.argtypes = [C.POINTER(ForwardArgs)]
        lib.gemlite_hip_forward.argtypes = [C.POINTER(ForwardArgs), C.c_void_p]
        lib.gemlite_hip_workspace_bytes.restype = C.c_uint64
        lib.gemlite_hip_workspace_bytes.argtypes = [C.POINTER(ForwardArgs)]
        lib.gemlite_hip_kernel_name.restype = C.c_char_p
        lib.gemlite_hip_kernel_name.argtypes = [C.POINTER(ForwardArgs)]
        lib.gemlite_hip_set_profile_events.restype = None
        lib.gemlite_hip_set_profile_events.argtypes = [C.c_void_p, C.c_void_p]
        lib.gemlite_hip_launch_noop.restype = C.c_int
        lib.gemlite_hip_launch_noop.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
        lib.gemlite_hip_scale_activations_per_token.restype = C.c_int
        lib.gemlite_hip_scale_activations_per_token.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        for name in ("gemlite_hip_scale_activations_mxfp8", "gemlite_hip_scale_activations_mxfp4",
                     "gemlite_hip_scale_activations_nvfp4"):
            getattr(lib, name).restype = C.c_int
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                           C.c_void_p]
        lib.gemlite_hip_pack_over_cols.restype = C.c_int
        lib.gemlite_hip_pack_over_cols.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        lib.gemlite_hip_unpack_over_cols.restype = C.c_int
        lib.gemlite_hip_unpack_over_cols.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        if lib.gemlite_hip_abi_version() != ABI_VERSION:
            raise GemliteHipError("libgemlite_hip.so ABI version mismatch; rebuild it")
        _lib = lib
    return _lib


EXPORTED_SYMBOLS = (
    "gemlite_hip_abi_version", "gemlite_hip_build_info", "gemlite_hip_status_string", "gemlite_hip_last_hip_error",
    "gemlite_hip_query", "gemlite_hip_workspace_bytes", "gemlite_hip_forward", "gemlite_hip_kernel_name",
    "gemlite_hip_set_profile_events", "gemlite_hip_launch_noop", "gemlite_hip_scale_activations_per_token",
    "gemlite_hip_scale_activations_mxfp8", "gemlite_hip_scale_activations_mxfp4", "gemlite_hip_scale_activations_nvfp4",
    "gemlite_hip_pack_over_cols",
    "gemlite_hip_unpack_over_cols",
)


def status_string(code: int) -> str:
    return load().gemlite_hip_status_string(int(code)).decode()


def raise_for_status(code: int, what: str):
    """Map a C status to the exception class the reference raises in the same situation."""
    if code == OK:
        return
    msg = f"{what}: {status_string(code)}"
    if code == ERR_UNSUPPORTED or code == ERR_BAD_SHAPE:
        raise NotImplementedError(msg)
    if code == ERR_LAUNCH:
        msg += f" (hipError_t={load().gemlite_hip_last_hip_error()})"
    err = GemliteHipError(msg)
    err.status = code
    raise err


def require_gpu_tensor(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise GemliteHipError(
            f"gemlite_amd: `{name}` lives on {t.device}; the HIP backend only runs on an MI355X device "
            "(there is deliberately no CPU fallback)")


# ------------------------------------------------------------------------------------------------------
# split-K workspace: one zero-initialised buffer per (device, stream); the kernels leave it zeroed again
# ------------------------------------------------------------------------------------------------------
_workspaces: dict = {}
_ws_lock = threading.Lock()


def workspace(device: torch.device, stream_handle: int, nbytes: int) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(stream_handle))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        with _ws_lock:
            ws = _workspaces.get(key)
            if ws is None or ws.numel() < nbytes:
                size = max(int(nbytes), 1 << 20)
                size = 1 << (size - 1).bit_length()  # grow geometrically
                ws = torch.zeros(size, dtype=torch.uint8, device=device)
                _workspaces[key] = ws
    return ws


class on_device:
    """Context for every ctypes launch: the library launches on the thread's CURRENT device, the tensors may live on another one
    (device_map / model-parallel set-ups move tensors without torch.cuda.set_device).  No-op when they agree."""
    __slots__ = ("_guard",)

    def __init__(self, device: torch.device):
        idx = device.index
        self._guard = torch.cuda.device(idx) if (idx is not None and idx != torch.cuda.current_device()) else None

    def __enter__(self):
        if self._guard is not None:
            self._guard.__enter__()
        return self

    def __exit__(self, *exc):
        if self._guard is not None:
            self._guard.__exit__(*exc)
        return False


def current_stream_handle(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device`, as an int."""
    return torch.cuda.current_stream(device).cuda_stream  # public API only (the raw private getter saved ~1 us)
