"""Decode tests/golden/pack_forward.npz (written by oracle/gen_golden.py from the reference)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CFG_FIELDS = ("nb", "gs", "N", "K", "in_dt", "out_dt", "fma", "scaled_act", "pb", "tdt", "zeros_kind", "scales_kind")
_TORCH = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16, 3: torch.float8_e4m3fn, 4: torch.int8, 6: torch.int32}


def as_torch(arr: np.ndarray, code: int) -> torch.Tensor:
    """Fixture array -> tensor of dtype code (bf16 travels as int16 bits, fp8 as uint8 bits)."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if code == 2 and t.dtype == torch.int16:
        return t.view(torch.bfloat16)
    if code == 3 and t.dtype == torch.uint8:
        return t.view(torch.float8_e4m3fn)
    return t


def load_cases():
    z = np.load(os.path.join(GOLDEN, "pack_forward.npz"))
    out = []
    for nm in z["names"]:
        nm = str(nm)
        cfg = dict(zip(CFG_FIELDS, (int(v) for v in z[nm + "/cfg"])))
        c = dict(name=nm, cfg=cfg, meta_args=[int(v) for v in z[nm + "/meta_args"]])
        for k in ("W_in", "scales_in", "zeros_in", "W_q", "W_q_stride", "scales", "zeros"):
            c[k] = z[f"{nm}/{k}"]
        c["x"], c["y"] = {}, {}
        for key in z.files:
            if key.startswith(nm + "/x_M"):
                c["x"][int(key.split("_M")[1])] = z[key]
            if key.startswith(nm + "/y_"):
                mt, M = key[len(nm) + 3:].rsplit("_M", 1)
                c["y"][(mt, int(M))] = z[key]
        out.append(c)
    return out
