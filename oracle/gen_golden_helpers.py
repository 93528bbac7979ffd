#!/usr/bin/env python3
"""Generate tests/golden/helpers.npz: what the REFERENCE's layer processors (gemlite/helper.py) produce on seeded inputs.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  The reference is imported on CPU with the same
process-local device fakes as oracle/gen_golden.py; its processors only run host code (`pack()` on CPU tensors), so
every output is the reference's own: packed / transposed W_q, scales, zeros, bias, the 12 `meta_args` ints.
hqq-object entry points are not exercised (the hqq package is not in this image); fp8 processors are given
`torch.float8_e4m3fn` explicitly because the reference's HIP default is the MI300X format e4m3fnuz.

Usage:  python oracle/gen_golden_helpers.py [--out tests/golden]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # sets TRITON_INTERPRET and provides the device fakes

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    G._fake_device_and_import()
    from gemlite import helper as H

    g = torch.Generator().manual_seed(4321)
    N, K = 48, 256
    W = (torch.randn(N, K, generator=g) / 30).to(torch.float16)
    bias = (torch.randn(N, generator=g) / 10).to(torch.float16)
    W_q4 = torch.randint(0, 16, (N, K), generator=g, dtype=torch.int32).to(torch.uint8)
    s64 = (torch.rand(N * K // 64, 1, generator=g) * 0.01 + 0.001).to(torch.float16)
    z64 = (torch.rand(N * K // 64, 1, generator=g) * 15).to(torch.float16)
    sch = (torch.rand(N, 1, generator=g) * 0.01 + 0.001).to(torch.float16)
    zch = (torch.rand(N, 1, generator=g) * 15).to(torch.float16)
    Wt = torch.randint(-1, 2, (N, K), generator=g).to(torch.float16)
    wscale = torch.tensor(0.0173)
    fp8 = torch.float8_e4m3fn

    cases = {
        "a16w8_int8_pre": lambda: H.A16W8(device="cpu").from_weights(W.clone(), bias.clone()),
        "a16w8_int8_post": lambda: H.A16W8(device="cpu", post_scale=True).from_weights(W.clone()),
        "a16w8_fp8_pre": lambda: H.A16W8(device="cpu", fp8=fp8).from_weights(W.clone()),
        "a16wn_g64": lambda: H.A16Wn(device="cpu").from_weights(W_q4.clone(), s64.clone(), z64.clone(), 4, 64, bias.clone()),
        "a16wn_channel": lambda: H.A16Wn(device="cpu", post_scale=True).from_weights(W_q4.clone(), sch.clone(), zch.clone(), 4, K),
        "a8w8_int8_dyn": lambda: H.A8W8_dynamic(device="cpu", fp8=False).from_weights(W.clone(), bias.clone()),
        "a8w8_fp8_dyn": lambda: H.A8W8_dynamic(device="cpu", fp8=fp8).from_weights(W.clone()),
        "a8w4_dyn_g64": lambda: H.A8Wn_HQQ_INT_dynamic(device="cpu", fp8=fp8, W_nbits=4).from_weights(W_q4.clone(), s64.clone(), z64.clone()),
        "a8w4_dyn_channel": lambda: H.A8Wn_HQQ_INT_dynamic(device="cpu", fp8=fp8, W_nbits=4, post_scale=True).from_weights(W_q4.clone(), sch.clone(), zch.clone()),
        "a16w158": lambda: H.A16W158_INT(device="cpu").from_weights(Wt.clone(), wscale, bias.clone()),
        "a8w158_dyn": lambda: H.A8W158_INT_dynamic(device="cpu").from_weights(Wt.clone(), wscale),
    }
    blob = dict(in_W=G._np(W), in_bias=G._np(bias), in_W_q4=G._np(W_q4), in_s64=G._np(s64), in_z64=G._np(z64),
                in_sch=G._np(sch), in_zch=G._np(zch), in_Wt=G._np(Wt), in_wscale=np.float32(wscale.item()))
    names = []
    for name, make in cases.items():
        lin = make()
        names.append(name)
        blob[f"{name}__W_q"] = G._np(lin.W_q.data)
        blob[f"{name}__W_q_dtype"] = np.array(str(lin.W_q.dtype))
        blob[f"{name}__W_q_strides"] = np.array(lin.W_q.stride(), np.int64)
        blob[f"{name}__scales"] = G._np(lin.scales.data.float())
        blob[f"{name}__scales_dtype"] = np.array(str(lin.scales.dtype))
        blob[f"{name}__zeros"] = G._np(lin.zeros.data.float())
        blob[f"{name}__zeros_dtype"] = np.array(str(lin.zeros.dtype))
        blob[f"{name}__bias"] = G._np(lin.bias.data.float()) if lin.bias is not None else np.zeros(0, np.float32)
        blob[f"{name}__meta"] = np.array(lin.get_meta_args(), np.int64)
        blob[f"{name}__modes"] = np.array([lin.W_group_mode, lin.channel_scale_mode], np.int64)
        print(name, lin.get_meta_args(), (lin.W_group_mode, lin.channel_scale_mode), tuple(lin.W_q.shape), lin.W_q.dtype)
    blob["names"] = np.array(names)
    np.savez_compressed(os.path.join(os.path.abspath(args.out), "helpers.npz"), **blob)
    print("helpers:", len(names), "cases")


if __name__ == "__main__":
    main()


def host_tables(out_dir):
    """Reference host-side lookup functions on a dense range: get_closest_m (autotune M buckets,
    triton_kernels/utils.py:136-174), get_matmul_type / get_default_gemv (core.py:100-114)."""
    G._fake_device_and_import()
    from gemlite import core as C
    from gemlite.triton_kernels import utils as U
    Ms = np.arange(0, 4300, dtype=np.int64)
    closest = np.array([U.get_closest_m(int(m)) for m in Ms], np.int64)
    kinds = sorted({C.get_matmul_type(int(m), nb) for m in (1, 2, 64, 65) for nb in (1, 2, 4, 8)})
    table = {}
    for nb in (1, 2, 4, 8):
        table[f"matmul_type_w{nb}"] = np.array([kinds.index(C.get_matmul_type(int(m), nb)) for m in Ms[1:200]], np.int64)
    np.savez_compressed(os.path.join(out_dir, "host_tables.npz"), Ms=Ms, closest_m=closest, kinds=np.array(kinds), **table)
    print("host_tables: closest_m for", len(Ms), "values;", kinds)


CTOR_GRID = dict(W_nbits=(1, 2, 3, 4, 8, 16), group_size=(None, 8, 16, 32, 64, 100, 128), in_features=(64, 96, 100, 4096),
                 dtype=("FP32", "FP16", "BF16", "FP8", "INT8", "FP8e5", "UINT8"))


def ctor_grid(out_dir):
    """Outcome of GemLiteLinear.__init__ (core.py:231-299) over a grid: exception class or the derived attributes."""
    import itertools
    G._fake_device_and_import()
    from gemlite import DType, GemLiteLinear
    rows = []
    for nb, gs, k, dt in itertools.product(*CTOR_GRID.values()):
        try:
            lin = GemLiteLinear(nb, gs, k, 64, getattr(DType, dt), DType.FP16, scaled_activations=True)
            rows.append(f"ok|{lin.group_size}|{lin.unpack_mask}|{int(lin.scaled_activations)}|{lin.acc_dtype.value}|{lin.meta_dtype.value}")
        except Exception as e:  # noqa: BLE001 - the class name is the recorded behaviour
            rows.append(type(e).__name__)
    np.savez_compressed(os.path.join(out_dir, "ctor_grid.npz"), rows=np.array(rows))
    print("ctor_grid:", len(rows), "combinations,", sum(r.startswith("ok") for r in rows), "accepted")


PACK_GRID = dict(nbits=(4, 2, 8), scales=("none", "group", "channel"), zeros=("none", "int", "tensor"), fma=(True, False),
                 in_dt=("FP16", "BF16", "INT8", "FP8"), scaled=(False, True))


def pack_inputs(nbits, scales_kind, zeros_kind, N=16, K=128, gs=64):
    """Seeded raw inputs of one pack() grid point (shared by the generator and the test)."""
    g = torch.Generator().manual_seed(1000 + nbits)
    W_q = torch.randint(0, 2 ** nbits, (N, K), generator=g, dtype=torch.int32).to(torch.uint8)
    n_groups = {"group": N * K // gs, "channel": N}.get(scales_kind, N * K // gs)
    scales = None if scales_kind == "none" else (torch.rand(n_groups, 1, generator=g) * 0.01 + 0.001).to(torch.float16)
    zeros = {"none": None, "int": 2 ** (nbits - 1)}.get(zeros_kind, "t")
    if zeros == "t":
        zeros = torch.round(torch.rand(n_groups, 1, generator=g) * (2 ** nbits - 1)).to(torch.float16)
    return W_q, scales, zeros


def pack_grid(out_dir):
    """What pack() decides (core.py:336-519) over a grid of scale / zero kinds, fma mode, input dtype and activation
    scaling: modes, meta_args, dtypes and shapes of the stored tensors, or the exception class."""
    import itertools
    G._fake_device_and_import()
    from gemlite import DType, GemLiteLinear
    rows = []
    for nb, sk, zk, fma, dt, sa in itertools.product(*PACK_GRID.values()):
        W_q, scales, zeros = pack_inputs(nb, sk, zk)
        gs = 128 if sk == "channel" else 64
        try:
            lin = GemLiteLinear(nb, gs, 128, 16, getattr(DType, dt), DType.FP16, scaled_activations=sa)
            lin.pack(W_q, scales, zeros, None, fma_mode=fma)
            rows.append("|".join(str(v) for v in (
                "ok", lin.W_group_mode, lin.channel_scale_mode, lin.get_meta_args(), str(lin.scales.dtype), tuple(lin.scales.shape),
                str(lin.zeros.dtype), tuple(lin.zeros.shape), float(lin.zeros.float().sum()), float(lin.scales.float().sum()))))
        except Exception as e:  # noqa: BLE001
            rows.append(type(e).__name__)
    np.savez_compressed(os.path.join(out_dir, "pack_grid.npz"), rows=np.array(rows))
    print("pack_grid:", len(rows), "combinations,", sum(r.startswith("ok") for r in rows), "accepted")


def dtype_table_and_state_dicts(out_dir):
    """DType members (dtypes.py:8-29) and the state_dict() of three packed reference layers (core.py:301-333, 503-517)."""
    G._fake_device_and_import()
    from gemlite import DType, GemLiteLinear
    from gemlite import helper as H
    blob = {"dtype_names": np.array([d.name for d in DType]), "dtype_values": np.array([d.value for d in DType], np.int64)}
    g = torch.Generator().manual_seed(77)
    W_q = torch.randint(0, 16, (32, 256), generator=g, dtype=torch.int32).to(torch.uint8)
    sc = (torch.rand(32 * 256 // 128, 1, generator=g) * 0.01 + 0.001).to(torch.float16)
    zr = torch.round(torch.rand(32 * 256 // 128, 1, generator=g) * 15).to(torch.float16)
    bias = (torch.randn(32, generator=g) / 10).to(torch.float16)
    W = (torch.randn(32, 256, generator=g) / 30).to(torch.float16)
    lin_a = GemLiteLinear(4, 128, 256, 32, DType.FP16, DType.FP16)
    lin_a.pack(W_q, sc, zr, bias)
    layers = {"a16w4": lin_a, "a8w8": H.A8W8_dynamic(device="cpu", fp8=False).from_weights(W.clone()),
              "bitnet": H.A16W158_INT(device="cpu").from_weights(torch.randint(-1, 2, (32, 256), generator=g).to(torch.float16), torch.tensor(0.03))}
    blob.update(sd_in_W_q=G._np(W_q), sd_in_scales=G._np(sc), sd_in_zeros=G._np(zr), sd_in_bias=G._np(bias))
    for name, lin in layers.items():
        sd = lin.state_dict()
        blob[f"sd_{name}__keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            blob[f"sd_{name}__{k}"] = G._np(v)
            blob[f"sd_{name}__{k}__dtype"] = np.array(str(v.dtype))
        blob[f"sd_{name}__meta_args"] = np.array(lin.get_meta_args(), np.int64)
        print(name, list(sd.keys()), lin.get_meta_args())
    np.savez_compressed(os.path.join(out_dir, "state_dicts.npz"), **blob)


if __name__ == "__main__" and os.environ.get("GEN_HOST_TABLES", "1") == "1":
    _out = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    host_tables(_out)
    ctor_grid(_out)
    pack_grid(_out)
    dtype_table_and_state_dicts(_out)
