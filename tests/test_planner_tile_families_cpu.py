"""End of round 6: the plans of the tile families that were re-fitted against every forced tile form (profiles/r06/scan_*.log) are frozen in
tests/golden/planner_tile_families.json — kernel name AND launch grid (tiles x K slices) per (family, M, N, K) over the 20 LLM layer shapes x M in {128, 256, 384, 512} —
so that a rule edited for one shape does not silently move the others, and the two properties those scans were about hold by construction:
the LDS-fed 128- / 256-row tiles never take more blocks than the next whole round of CUs needs while half the chip is already busy.
Regenerate after an intended change: `python tests/test_planner_tile_families_cpu.py --write`."""
import ctypes as C
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemlite_amd import _hip  # noqa: E402
from tests.test_host_cpu import _args  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "planner_tile_families.json")
SHAPES = [(1024, 4096), (1536, 8960), (2048, 8192), (2560, 9728), (3072, 8192), (4096, 1024), (4096, 4096), (4096, 11008), (4096, 14336), (5120, 5120), (5120, 13824), (6144, 4096),
          (8192, 2048), (8192, 3072), (8192, 8192), (8960, 1536), (11008, 4096), (12288, 4096), (13824, 5120), (14336, 4096)]
MS = (128, 256, 384, 512)


def _mx(in_dt, nbits, M, c_mode, N, K, group=32):
    """A block-scaled request in the K-contiguous layout of pack() (input types: 14 / 15 = fp16 / bf16 over MX weights, 16 = MXFP8, 17 = MXFP4, 18 = NVFP4)."""
    a = _hip.ForwardArgs()
    a.struct_size = C.sizeof(_hip.ForwardArgs)
    a.matmul_type = -1
    a.x = a.w_q = a.scales = a.out = a.scales_x = 0x1000
    a.M, a.N, a.K = M, N, K
    a.W_nbits, a.group_size, a.unpack_mask = nbits, group, 2 ** nbits - 1
    a.elements_per_sample = 1 if nbits == 8 else 2
    a.w_pack_bits = 0 if nbits == 8 else 8
    a.w_dtype = 3 if nbits == 8 else 5
    a.input_dtype, a.output_dtype, a.meta_dtype = in_dt, (1 if in_dt == 14 else 2), 5
    a.channel_scale_mode, a.W_group_mode = c_mode, 0
    a.stride_xm, a.stride_xk = (K if in_dt in (14, 15, 16) else K // 2), 1
    a.stride_wk, a.stride_wn = 1, (K if nbits == 8 else K // 2)
    a.stride_om, a.stride_on = N, 1
    a.stride_meta_g, a.stride_meta_n = N, 1
    a.stride_sx_m = K // group
    return a


def _a8(in_dt, M, N, K):
    a = _args(M=M, N=N, K=K, nbits=8, e=1, in_dt=in_dt, w_mode=0, c_mode=3, out_dt=1, gs=K)
    a.scales_x = 0x1000
    return a


FAMILIES = {
    "a8w8_int8": lambda M, N, K: _a8(4, M, N, K),
    "a8w8_fp8": lambda M, N, K: _a8(3, M, N, K),
    "a16w8_int8": lambda M, N, K: _args(M=M, N=N, K=K, nbits=8, e=1, in_dt=1, w_dtype=4, w_mode=2, gs=K),
    "a16w2": lambda M, N, K: _args(M=M, N=N, K=K, nbits=2, gs=128, in_dt=1),
    "a16w4": lambda M, N, K: _args(M=M, N=N, K=K, nbits=4, gs=128, in_dt=2),
    "mx_a8w8": lambda M, N, K: _mx(16, 8, M, 4, N, K),
    "mx_a8w4": lambda M, N, K: _mx(16, 4, M, 2, N, K),
    "mx_a4w4": lambda M, N, K: _mx(17, 4, M, 4, N, K),
    "a16w8_mxfp": lambda M, N, K: _mx(14, 8, M, 0, N, K),
    "a16w4_mxfp": lambda M, N, K: _mx(15, 4, M, 0, N, K),
}


def _plan(lib, a):
    name = lib.gemlite_hip_kernel_name(C.byref(a)).decode()
    ws = int(lib.gemlite_hip_workspace_bytes(C.byref(a)))
    return name, ws


def _rows():
    lib = _hip.load()
    lib.gemlite_hip_workspace_bytes.restype = C.c_uint64
    out = []
    for fam, mk in FAMILIES.items():
        for M in MS:
            for (N, K) in SHAPES:
                if fam == "mx_a4w4" and K % 64 != 0:
                    continue
                name, ws = _plan(lib, mk(M, N, K))
                out.append([fam, M, N, K, name, ws])
    return out


def test_tile_family_plans_are_frozen():
    fx = json.load(open(GOLDEN))["rows"]
    assert len(fx) >= 10 * 4 * 20 - 40
    got = {tuple(r[:4]): (r[4], r[5]) for r in _rows()}
    moved = [(tuple(r[:4]), (r[4], r[5]), got.get(tuple(r[:4]))) for r in fx if got.get(tuple(r[:4])) != (r[4], r[5])]
    assert not moved, moved[:8]
    assert not [r for r in fx if "generic" in r[4]], "a tile-family shape on the coverage kernel"


def test_lds_fed_tiles_do_not_spill_their_k_slices_into_a_second_round():
    """The slab bytes of a plan give tiles x slices (slab = tiles x slices x tile rows x 128 x 4 bytes): for the one-block-per-CU tiles (A8W8 `lds`, A16W8,
    A16W2 at 128 / 256 rows) no plan may hold between 257 and 511 blocks when half as many slices would already have kept 128 CUs busy."""
    fx = json.load(open(GOLDEN))["rows"]
    checked = 0
    for fam, M, N, K, name, ws in fx:
        if fam not in ("a8w8_int8", "a8w8_fp8", "a16w8_int8", "a16w2") or ws == 0:
            continue
        if not ("lds_kernel" in name or "a16w8_kernel<" in name or "w2_mma_kernel<" in name):
            continue
        rows = {"<128x128>": 128, "<256x128>": 256, "<64x128>": 64, "<32x128>": 32}.get(name[name.index("<"):])
        if rows is None or rows < 128:
            continue
        tiles = (N // 128) * ((M + rows - 1) // rows)
        slab = ws - 262144   # (the ticket counters in front of the slabs: COUNTER_BYTES, gl_common.h)
        blocks = slab // (rows * 128 * 4)
        sk = max(1, round(blocks / tiles))
        checked += 1
        if sk > 1 and tiles * sk > 256:
            assert tiles * (sk - 1) < 128 or tiles * sk >= 448, (fam, M, N, K, name, tiles, sk)
    assert checked >= 40


if __name__ == "__main__" and "--write" in sys.argv:
    rows = _rows()
    with open(GOLDEN, "w") as f:
        f.write('{"_note": "kernel name and workspace bytes (= counter page + tiles x K slices x tile bytes) the C ABI plans for the re-fitted tile families over 20 LLM layer shapes x M = 128 / 256 / 384 / 512 '
                '(end of round 6, profiles/r06/scan_*.log); regenerate with python tests/test_planner_tile_families_cpu.py --write", "rows": [\n')
        f.write(",\n".join(" " + json.dumps(r) for r in rows))
        f.write("\n]}\n")
    print(len(rows), "rows")
