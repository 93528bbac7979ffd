"""Pin the CPU oracle to the reference: every fixture in tests/golden/ was produced by the
reference's own code (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import gemlite_oracle as O
from tests.golden_util import GOLDEN, as_torch, load_cases

CASES = load_cases()


def test_bitpack_matches_reference_bits():
    z = np.load(os.path.join(GOLDEN, "bitpack.npz"))
    n = 0
    for nb in (1, 2, 4, 8):
        for pb in (8, 16, 32):
            if f"in_{nb}_{pb}" not in z.files:
                continue
            W, ref = z[f"in_{nb}_{pb}"], z[f"out_{nb}_{pb}"]
            assert int(z[f"e_{nb}_{pb}"]) == pb // nb
            mine = O.pack_over_cols(W, nb, pb)
            assert mine.dtype == ref.dtype and mine.shape == ref.shape
            assert np.array_equal(mine, ref), f"pack mismatch nbits={nb} pack={pb}"
            assert np.array_equal(O.unpack_over_cols(ref, nb, pb), W)
            n += 1
    assert n == 12


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_pack_modes_and_layout(case):
    """W_group_mode / channel_scale_mode / metadata layout == the reference's pack()."""
    cfg, meta = case["cfg"], case["meta_args"]
    zeros_kind = {0: "none", 1: "int", 2: "tensor"}[cfg["zeros_kind"]]
    has_scales = cfg["scales_kind"] != 0
    wgm, csm, folded = O.resolve_modes(has_scales=has_scales, scales_numel=case["scales_in"].size,
                                       zeros_kind=zeros_kind, out_features=cfg["N"],
                                       scaled_activations=bool(cfg["scaled_act"]), fma_mode=bool(cfg["fma"]))
    assert (wgm, csm) == (meta[10], meta[9])
    tdt = cfg["tdt"]
    if meta[4] > 1:  # packed
        pb = 32 if cfg["pb"] < 0 else cfg["pb"]
        assert meta[4] == pb // cfg["nb"]
        assert np.array_equal(O.pack_over_cols(case["W_in"], cfg["nb"], pb), case["W_q"])
    scales_in = O.to_f64(as_torch(case["scales_in"], tdt)) if has_scales else None
    zeros_in = O.to_f64(as_torch(case["zeros_in"], tdt)) if zeros_kind == "tensor" else (
        int(case["zeros_in"][0]) if zeros_kind == "int" else None)
    s_l, z_l = O.layout_meta(scales_in, zeros_in, cfg["N"], folded, tdt)
    if has_scales:
        assert np.array_equal(s_l, O.to_f64(as_torch(case["scales"], tdt)))
    if zeros_kind == "tensor":
        assert np.array_equal(z_l, O.to_f64(as_torch(case["zeros"], tdt))), "zeros layout / fma folding differs"
    elif zeros_kind == "int":
        assert case["zeros"].size == 1 and int(case["zeros"].reshape(-1)[0]) == int(zeros_in)


def _oracle_forward(case, M, mt, faithful):
    cfg, meta = case["cfg"], case["meta_args"]
    (scaled_act, nb, gs, _mask, e, in_dt, out_dt, _acc, meta_dt, csm, wgm, _contig) = meta
    tdt = cfg["tdt"]
    x = as_torch(case["x"][M], in_dt if cfg["scaled_act"] == 0 else tdt)
    scales_x = None
    xf = O.to_f64(x)
    if scaled_act:
        xf, scales_x = O.scale_activations_per_token(x, in_dt)
    s = O.to_f64(as_torch(case["scales"], meta_dt)) if case["scales"].size else None
    zk = cfg["zeros_kind"]
    z = None
    if zk == 2:
        z = O.to_f64(as_torch(case["zeros"], meta_dt))
    elif zk == 1:
        z = case["zeros"].astype(np.float64).reshape(-1)
    mc = meta_dt if faithful else None
    if e > 1:
        pb = 32 if cfg["pb"] < 0 else cfg["pb"]
        return O.forward_packed(xf, case["W_q"], s, z, W_nbits=nb, group_size=gs, W_group_mode=wgm,
                                channel_scale_mode=csm, scales_x=scales_x, zero_is_scalar=(zk == 1), pack_bits=pb,
                                meta_code=mc, output_code=out_dt)
    w_code = 1 if nb == 16 else (4 if in_dt == 4 else 3)  # unpacked weights: fp16 / int8 / fp8e4m3
    W_kn = O.to_f64(as_torch(case["W_in"], w_code)).T  # W_q = W.t()  (core.py:377)
    W = O.dequantize(W_kn, s if wgm >= 2 else None, z if wgm in (1, 3, 4) else None, gs, wgm, zk == 1, mc)
    return O.forward(xf, W, scales_w_channel=s if csm in (1, 3) else None, scales_x=scales_x,
                     channel_scale_mode=csm, meta_code=mc, output_code=out_dt)


FWD = [(c, M, mt) for c in CASES for (mt, M) in sorted(c["y"])]


@pytest.mark.parametrize("case,M,mt", FWD, ids=[f"{c['name']}-{mt}-M{M}" for c, M, mt in FWD])
def test_forward_matches_reference_kernels(case, M, mt):
    """Oracle vs outputs of the reference's own Triton kernels (interpreter).  The exact oracle
    must sit within the reference's accumulation noise; bounds are relative to mean|y|."""
    y_ref = case["y"][(mt, M)].astype(np.float64)
    y = _oracle_forward(case, M, mt, faithful=False)
    assert y.shape == y_ref.shape
    scale = max(np.abs(y_ref).mean(), 1e-6)
    err = np.abs(y - y_ref)
    # GEMV_REVSPLITK accumulates in fp16/bf16 (gemv_revsplitK_kernels.py:426-430); GEMM_SPLITK in fp32.
    bf16 = case["meta_args"][5] == 2 or case["meta_args"][8] == 2
    tol_mean = (4e-3 if mt == "GEMV_REVSPLITK" else 1.5e-3) * (8 if bf16 else 1)
    assert err.mean() / scale < tol_mean, (err.mean(), scale)
    assert err.max() / scale < 25 * tol_mean, (err.max(), scale)


def test_act_quant_matches_reference():
    z = np.load(os.path.join(GOLDEN, "act_quant.npz"))
    for tag, code in (("int8", O.INT8), ("fp8e4", O.FP8E4), ("fp8e5", O.FP8E5)):
        for M, K in ((1, 256), (5, 384)):
            x = z[f"{tag}_x_{M}_{K}"]
            q_ref, s_ref = z[f"{tag}_q_{M}_{K}"], z[f"{tag}_s_{M}_{K}"]
            q, s = O.scale_activations_per_token(torch.from_numpy(x), code)
            assert np.array_equal(s.astype(np.float32), s_ref.astype(np.float32)), tag
            assert np.array_equal(q, q_ref.astype(np.float64)), tag


def test_c_oracle_matches_numpy_oracle():
    """The plain-C restatement (oracle/oracle.c) agrees with the numpy oracle on every packed golden case."""
    import ctypes
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True)
    lib = ctypes.CDLL(so)
    fp, ip, dp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)
    lib.oracle_forward_packed.restype = ctypes.c_int
    lib.oracle_forward_packed.argtypes = [fp, ip, fp, fp, fp, fp, dp] + [ctypes.c_int64] * 3 + [ctypes.c_int] * 6
    n_checked = 0
    for case in CASES:
        meta, cfg = case["meta_args"], case["cfg"]
        (_sa, nb, gs, _mask, e, in_dt, out_dt, _acc, meta_dt, csm, wgm, _c) = meta
        if e == 1 or cfg["scaled_act"] or case["W_q"].dtype != np.int32:
            continue
        M = sorted(case["x"])[0]
        x = O.to_f64(as_torch(case["x"][M], in_dt)).astype(np.float32)
        s = O.to_f64(as_torch(case["scales"], meta_dt)).astype(np.float32) if case["scales"].size else np.zeros(1, np.float32)
        zk = cfg["zeros_kind"]
        z = (O.to_f64(as_torch(case["zeros"], meta_dt)) if zk == 2 else case["zeros"].astype(np.float64).reshape(-1)
             if zk == 1 else np.zeros(1)).astype(np.float32)
        K, N = cfg["K"], cfg["N"]
        y = np.zeros((M, N), np.float64)
        w = np.ascontiguousarray(case["W_q"])
        eff_gs = gs if (wgm >= 2 or (wgm == 1 and zk == 2 and csm not in (1, 3))) else K
        rc = lib.oracle_forward_packed(x.ctypes.data_as(fp), w.ctypes.data_as(ip), np.ascontiguousarray(s).ctypes.data_as(fp),
                                       np.ascontiguousarray(z).ctypes.data_as(fp),
                                       np.ascontiguousarray(s.reshape(-1)).ctypes.data_as(fp), None, y.ctypes.data_as(dp),
                                       M, N, K, nb, 32, eff_gs, wgm, csm, int(zk == 1))
        assert rc == 0
        y_np = O.forward_packed(x.astype(np.float64), case["W_q"], O.to_f64(as_torch(case["scales"], meta_dt)) if case["scales"].size else None,
                                z.astype(np.float64) if zk else None, W_nbits=nb, group_size=gs, W_group_mode=wgm,
                                channel_scale_mode=csm, zero_is_scalar=(zk == 1))
        assert np.allclose(y, y_np, rtol=1e-9, atol=1e-12), case["name"]
        n_checked += 1
    assert n_checked >= 8
