"""Block-scaled formats (MXFP8 / MXFP4 / NVFP4), CPU side: the numpy oracle and the package's host code against
tests/golden/mx.npz = outputs of the reference itself (oracle/gen_golden_mx.py).  No GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import mx_oracle as MX
from tests.golden_util import GOLDEN

Z = np.load(os.path.join(GOLDEN, "mx.npz"))


def _bf16(arr):
    return torch.from_numpy(np.ascontiguousarray(arr)).view(torch.bfloat16)


def _x(tag):
    a = Z[f"act_{tag}_x"]
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.view(torch.bfloat16) if tag == "bf16" else t).float().numpy()


W_IN = _bf16(Z["wq_in_W"]).float().numpy()


# ------------------------------------------------------------------------------------------------ oracle vs reference
@pytest.mark.parametrize("name,fn", [
    ("mxfp8", lambda: MX.quantize_mxfp8(W_IN)), ("mxfp4", lambda: MX.quantize_mxfp4(W_IN)),
    ("mxfp4_w1", lambda: MX.quantize_mxfp4(W_IN, window_size=1)), ("nvfp4", lambda: MX.quantize_nvfp4(W_IN)),
    ("nvfp4_w1", lambda: MX.quantize_nvfp4(W_IN, window_size=1))])
def test_oracle_weight_quantiser_matches_the_reference(name, fn):
    q, s = fn()
    assert np.array_equal(s.reshape(-1), Z[f"wq_{name}_s"].reshape(-1)), "block scales differ"
    assert np.array_equal(q.reshape(-1), Z[f"wq_{name}_q"].reshape(-1)), "elements differ"


@pytest.mark.parametrize("tag", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["mxfp8", "mxfp4", "nvfp4"])
def test_oracle_activation_quantiser_matches_the_reference(tag, name):
    y, s = getattr(MX, "scale_activations_" + name)(_x(tag))
    ref_y, ref_s = Z[f"act_{tag}_{name}_y"], Z[f"act_{tag}_{name}_s"]
    assert y.shape == ref_y.shape and s.shape == ref_s.shape  # incl. the rows the reference pads M to
    assert np.array_equal(s, ref_s), "block scales differ"
    assert np.array_equal(y, ref_y), "elements differ"


def test_oracle_fp4_tables_and_fp8_codec_round_trip():
    allb = np.arange(256, dtype=np.uint8)
    vals = MX.fp8_e4m3_decode(allb)
    ok = ~np.isnan(vals)
    assert np.array_equal(MX.fp8_e4m3_encode(vals[ok]), allb[ok] if True else None) or np.array_equal(
        MX.fp8_e4m3_decode(MX.fp8_e4m3_encode(vals[ok])), vals[ok])
    t = torch.from_numpy(allb.copy()).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(t[ok], vals[ok])
    codes = np.arange(16, dtype=np.uint8)
    assert np.array_equal(MX.fp4_unpack(MX.fp4_pack_codes(codes[None, :]))[0], MX.FP4_VALUES)
    assert np.array_equal(MX.e8m0_decode(np.array([127, 128, 97], dtype=np.uint8)), [1.0, 2.0, 2.0 ** -30])


# ------------------------------------------------------------------------------------------------ package host code
def test_package_weight_quantiser_matches_the_reference():
    from gemlite_amd.quant_utils import WeightQuantizerMXFP
    wq = WeightQuantizerMXFP(compute_dtype=torch.bfloat16, device="cpu")
    W = _bf16(Z["wq_in_W"])
    for name, fn in (("mxfp8", lambda: wq.quantize_mxfp8(W, index=True)), ("mxfp4", lambda: wq.quantize_mxfp4(W, index=True)),
                     ("mxfp4_w1", lambda: wq.quantize_mxfp4(W, window_size=1, index=True)),
                     ("nvfp4", lambda: wq.quantize_nvfp4(W, index=True)),
                     ("nvfp4_w1", lambda: wq.quantize_nvfp4(W, window_size=1, index=True))):
        q, s = fn()
        assert np.array_equal(q.contiguous().view(torch.uint8).numpy().reshape(-1), Z[f"wq_{name}_q"].reshape(-1)), name
        assert np.array_equal(s.contiguous().view(torch.uint8).numpy().reshape(-1), Z[f"wq_{name}_s"].reshape(-1)), name
    # dequantize() undoes the scaling: values are e2m1 * 2^e exactly
    q, s = wq.quantize_mxfp4(W, index=True)
    back = wq.dequantize(q, s, shape=W.shape, dtype=torch.float32)
    assert (back - W.float()).abs().mean().item() < 0.02


PROCS = {
    "a16w8_mxfp": lambda H: H.A16W8_MXFP(device="cpu", dtype=torch.bfloat16),
    "a16w4_mxfp": lambda H: H.A16W4_MXFP(device="cpu", dtype=torch.float16),
    "a8w8_mxfp_dyn_post": lambda H: H.A8W8_MXFP_dynamic(device="cpu", dtype=torch.bfloat16, post_scale=True),
    "a8w8_mxfp_dyn_micro": lambda H: H.A8W8_MXFP_dynamic(device="cpu", dtype=torch.bfloat16, post_scale=False),
    "a8w4_mxfp_dyn": lambda H: H.A8W4_MXFP_dynamic(device="cpu", dtype=torch.bfloat16, post_scale=False),
    "a4w4_mxfp_dyn": lambda H: H.A4W4_MXFP_dynamic(device="cpu", dtype=torch.bfloat16),
    "a4w4_nvfp_dyn": lambda H: H.A4W4_NVFP_dynamic(device="cpu", dtype=torch.float16),
}


@pytest.mark.parametrize("name", list(PROCS))
def test_processors_pack_like_the_reference(name):
    """W_q / scales bytes, their shapes AND strides, the 12 meta ints and the bias of every MXFP / NVFP processor."""
    from gemlite_amd import helper as H
    W = _bf16(Z["wq_in_W"])
    lin = torch.nn.Linear(W.shape[1], W.shape[0], bias=True, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(W)
        lin.bias.copy_(_bf16(Z["proc_in_bias"]))
    layer = PROCS[name](H).from_linear(lin, del_orig=False)
    wq, sc = layer.W_q.data, layer.scales.data
    assert list(wq.shape) + list(wq.stride()) == [int(v) for v in Z[f"proc_{name}_W_q_shape_stride"]]
    assert list(sc.shape) + list(sc.stride()) == [int(v) for v in Z[f"proc_{name}_scales_shape_stride"]]
    assert np.array_equal(wq.contiguous().view(torch.uint8).numpy(), Z[f"proc_{name}_W_q"])
    assert np.array_equal(sc.contiguous().view(torch.uint8).numpy(), Z[f"proc_{name}_scales"])
    assert layer.get_meta_args() == [int(v) for v in Z[f"proc_{name}_meta"]]
    ref_bias = Z[f"proc_{name}_bias"]
    mine = layer.bias.data
    mine = mine.view(torch.int16).numpy() if mine.dtype == torch.bfloat16 else mine.numpy()
    assert np.array_equal(mine, ref_bias)


def test_block_scaled_layers_refuse_cpu_tensors():
    from gemlite_amd import _hip
    from gemlite_amd import helper as H
    lin = torch.nn.Linear(256, 128, bias=False, dtype=torch.bfloat16)
    layer = H.A4W4_MXFP_dynamic(device="cpu", dtype=torch.bfloat16).from_linear(lin, del_orig=False)
    with pytest.raises(_hip.GemliteHipError):
        layer(torch.randn(2, 256, dtype=torch.bfloat16))


@pytest.mark.parametrize("fam", ["mxfp8", "mxfp8_w4", "mxfp4", "a16_mxw8", "a16_mxw4", "nvfp4"])
def test_block_scaled_few_row_to_tile_hand_over_is_monotone(fam):
    """Round 4, late: walking M upwards on the four layer sizes of the sweep, every block-scaled family hands over ONCE from
    its few-row kernel to a tile kernel, never reaches the coverage kernel, and has left the few-row kernels by 65 rows."""
    import ctypes as C
    from gemlite_amd import _hip
    lib = _hip.load()
    buf = (C.c_uint8 * 64)()
    ptr = C.addressof(buf) // 16 * 16 + 16
    in_dt, nbits, c_mode, group = {"mxfp8": (16, 8, 4, 32), "mxfp8_w4": (16, 4, 2, 32), "mxfp4": (17, 4, 4, 32), "a16_mxw8": (15, 8, 0, 32),
                                   "a16_mxw4": (14, 4, 0, 32), "nvfp4": (18, 4, 4, 16)}[fam]
    for N, K in ((4096, 4096), (8192, 8192), (4096, 14336), (14336, 4096)):
        tiled_at = None
        for M in list(range(1, 66)) + [96, 128, 256, 512]:
            a = _hip.ForwardArgs()
            a.struct_size = C.sizeof(_hip.ForwardArgs)
            a.matmul_type = -1
            a.x = a.w_q = a.scales = a.out = a.scales_x = ptr
            a.M, a.N, a.K = M, N, K
            a.W_nbits, a.group_size, a.unpack_mask = nbits, group, 2 ** nbits - 1
            a.elements_per_sample = 1 if nbits == 8 else 2
            a.w_pack_bits = 0 if nbits == 8 else 8
            a.w_dtype = 3 if nbits == 8 else 5
            a.input_dtype, a.output_dtype, a.meta_dtype = in_dt, (1 if in_dt in (14, 18) else 2), 5
            a.channel_scale_mode, a.W_group_mode = c_mode, 0
            a.stride_xm, a.stride_xk = (K // 2 if in_dt in (17, 18) else K), 1
            a.stride_wk, a.stride_wn = 1, (K if nbits == 8 else K // 2)
            a.stride_om, a.stride_on = N, 1
            a.stride_meta_g, a.stride_meta_n = N, 1
            a.stride_sx_m = K // group
            assert lib.gemlite_hip_query(C.byref(a)) == 0, (fam, N, K, M)
            name = lib.gemlite_hip_kernel_name(C.byref(a)).decode()
            assert "generic" not in name, (fam, N, K, M, name)
            few = "_rows_" in name or "gemv" in name
            if not few and tiled_at is None:
                tiled_at = M
            if tiled_at is not None:
                assert not few, (fam, N, K, M, name, "tile kernel since M = %d" % tiled_at)
        assert tiled_at is not None and tiled_at <= 65, (fam, N, K, tiled_at)


def test_c_abi_routes_block_scaled_formats():
    """gemlite_hip_kernel_name never launches: which kernel each format pair resolves to (K-contiguous layout of pack())."""
    import ctypes as C
    from gemlite_amd import _hip
    lib = _hip.load()
    buf = (C.c_uint8 * 64)()
    ptr = C.addressof(buf) // 16 * 16 + 16

    def args(in_dt, nbits, M, c_mode, N=4096, K=4096, group=32):
        a = _hip.ForwardArgs()
        a.struct_size = C.sizeof(_hip.ForwardArgs)
        a.matmul_type = -1
        a.x = a.w_q = a.scales = a.out = a.scales_x = ptr
        a.M, a.N, a.K = M, N, K
        a.W_nbits, a.group_size, a.unpack_mask = nbits, group, 2 ** nbits - 1
        a.elements_per_sample = 1 if nbits == 8 else 2
        a.w_pack_bits = 0 if nbits == 8 else 8
        a.w_dtype = 3 if nbits == 8 else 5
        a.input_dtype, a.output_dtype, a.meta_dtype = in_dt, 2, 5
        a.channel_scale_mode, a.W_group_mode = c_mode, 0
        xb = K if in_dt == 16 else (K // 2 if in_dt in (17, 18) else K)
        a.stride_xm, a.stride_xk = xb, 1
        a.stride_wk, a.stride_wn = 1, (K if nbits == 8 else K // 2)
        a.stride_om, a.stride_on = N, 1
        a.stride_meta_g, a.stride_meta_n = N, 1
        a.stride_sx_m = K // group
        return a

    name = lambda a: lib.gemlite_hip_kernel_name(C.byref(a)).decode()  # noqa: E731
    # round 4: 65 .. 384 rows (512 for fp4 x fp4 and one-round shapes) on 64 x 64 tiles with K unsplit
    assert name(args(16, 8, 256, 4, N=8192, K=8192)) == "gemm_mx_a8w8_sq_kernel<64x64>"
    assert name(args(16, 4, 256, 2, N=8192, K=8192)) == "gemm_mx_a8w4_sq_kernel<64x64>"
    assert name(args(17, 4, 256, 4, N=8192, K=8192)) == "gemm_mx_a4w4_sq_kernel<64x64>"
    assert name(args(16, 8, 65, 4)) == "gemm_mx_a8w8_sq_kernel<64x64>"
    assert name(args(16, 8, 512, 4)) == "gemm_mx_a8w8_sq_kernel<64x64>"    # 512 tiles, K <= 4096
    assert name(args(17, 4, 512, 4, N=8192, K=8192)) == "gemm_mx_a4w4_sq_kernel<64x64>"
    assert name(args(16, 4, 512, 2, N=8192, K=8192)) == "gemm_mx_a8w4_sq_kernel<64x64>"   # (late round 6: up to 512 rows everywhere — 77.8 -> 70.4 us)
    assert name(args(16, 8, 1024, 4)) == "gemm_mx_a8w8_sq_kernel<64x64>"                    # fp8 weights: up to 1024 rows (4096^2: 52.2 -> 35.2 us)
    assert name(args(16, 4, 1024, 2, N=8192, K=8192)) != "gemm_mx_a8w4_sq_kernel<64x64>"   # fp4 weights above 512 rows: only up to N K = 4096^2
    a = args(16, 8, 256, 4)
    a.tuning[0] = 2                                                       # A/B switch: the 128-column kernel with K slices
    assert name(a) == "gemm_mx_a8w8_kernel<128x128>"  # the tallest tile that fills the chip with <= K / 1024 slices
    a.tuning[0] = 6                                                       # ... and the forced form, at any M
    a.M = 700
    assert name(a) == "gemm_mx_a8w8_sq_kernel<64x64>"
    assert name(args(16, 8, 2048, 4, N=8192, K=8192)) == "gemm_mx_a8w8_tile_kernel<256x256>"  # >= 96 tiles of 256 x 256
    assert name(args(16, 8, 512, 4, N=8192, K=8192)) == "gemm_mx_a8w8_sq_kernel<64x64>"       # (late round 6: 94.0 -> 82.7 us; rounds 4-5: the 128-row kernel)
    # late round 6: a long K under 320 .. 640 of the 64 x 64 tiles goes back to the 128-row tiles with K slices (5120 x 13824 M = 512: 134.0 -> 104.4 us)
    assert name(args(16, 8, 512, 4, N=5120, K=13824)) == "gemm_mx_a8w8_kernel<128x128>"
    assert name(args(16, 8, 512, 4, N=2560, K=9728)) == "gemm_mx_a8w8_kernel<128x128>"      # 320 tiles: 0.625 of a round of 512
    assert name(args(16, 8, 384, 4, N=5120, K=13824)) == "gemm_mx_a8w8_sq_kernel<64x64>"    # 480 tiles: 0.94 (59.8 vs 76.1 us)
    assert name(args(16, 4, 256, 2, N=5120, K=13824)) == "gemm_mx_a8w4_kernel<128x128>"     # fp4 weights: only the largest layers
    assert name(args(16, 4, 512, 2, N=2560, K=9728)) == "gemm_mx_a8w4_sq_kernel<64x64>"
    assert name(args(17, 4, 512, 4, N=5120, K=13824)) == "gemm_mx_a4w4_sq_kernel<64x64>"    # fp4 x fp4 stays (65.1 vs 65.7)
    assert name(args(16, 8, 256, 4, N=4096, K=14336)) == "gemm_mx_a8w8_sq_kernel<64x64>"    # 256 tiles stay (38.0 vs 53.0)
    assert name(args(16, 8, 1, 4)) == "mx_rows_a8w8_kernel<16x16>"    # round 4: fp8 / fp4 activations take the few-row MFMA kernel from 1 row
    assert name(args(17, 4, 4, 4)) == "mx_rows_a4w4_kernel<16x16>"
    a = args(16, 8, 1, 4)
    a.tuning[0] = 5                                                       # A/B switch: the streaming kernel of rounds 2-3 (up to 4 rows)
    assert name(a) == "mx_gemv_w8_kernel"
    assert name(args(16, 8, 1, 4, K=4096 + 32)) == "mx_gemv_w8_kernel"   # K % 128 != 0
    assert name(args(14, 8, 1, 0)) == "a16w8_mxfp_rows_kernel<16x16>"    # round 4: 16-bit activations x MX weights, 1 .. 64 rows: the A16W8 rows kernel
    assert name(args(15, 4, 40, 0)) == "gemm_a16w4_mxfp_kernel<64x64>"   # round 4: rows only while M N K <= 570 M (250 M for layers > 32 M weights)
    assert name(args(15, 4, 30, 0)) == "a16w4_mxfp_rows_kernel<32x16>"
    assert name(args(15, 4, 8, 0, N=8192, K=8192)) == "gemm_a16w4_mxfp_kernel<32x128>"
    assert name(args(15, 4, 3, 0, N=8192, K=8192)) == "a16w4_mxfp_rows_kernel<16x16>"
    assert name(args(17, 4, 22, 4)) == "mx_rows_a4w4_kernel<32x16>"           # fp4 x fp4: rows up to 22 rows below 128 column tiles, 2 rows from there
    assert name(args(17, 4, 23, 4)) == "gemm_mx_a4w4_sq_kernel<64x64>"
    assert name(args(17, 4, 3, 4, N=8192, K=8192)) == "gemm_mx_a4w4_sq_kernel<64x64>"
    assert name(args(17, 4, 2, 4, N=8192, K=8192)) == "mx_rows_a4w4_kernel<16x16>"
    assert name(args(16, 8, 16, 4, N=8192, K=8192)) == "mx_rows_a8w8_kernel<16x16>"   # fp8 activations: M N K <= 1.1 G from 128 column tiles
    assert name(args(16, 8, 17, 4, N=8192, K=8192)) == "gemm_mx_a8w8_sq_kernel<64x64>"
    assert name(args(18, 4, 30, 4, group=16)) == "nvfp4_rows_kernel<32x16>"     # NVFP4: rows while M N K <= 600 M
    a = args(18, 4, 48, 4, group=16)
    a.out = ptr + 2                                                        # an output the tile kernels cannot store to (2-byte aligned): past the
    assert name(a) == "nvfp4_rows_kernel<64x16>"                           # budget the few-row kernel still takes it — not the coverage kernel
    a = args(14, 4, 48, 0)
    a.output_dtype, a.out = 1, ptr + 2
    assert name(a) == "a16w4_mxfp_rows_kernel<64x16>"
    assert name(args(14, 8, 1, 0, K=4096 + 32)) == "mx_gemv_w8_kernel"   # K % 64 != 0: the streaming kernel
    a = args(14, 8, 1, 0)
    a.tuning[0] = 5
    assert name(a) == "mx_gemv_w8_kernel"
    assert name(args(15, 8, 100, 0)) == "gemm_a16w8_mxfp_kernel<64x64>"     # above 64 rows: the tile kernel (bf16 x, bf16 out)
    assert name(args(17, 4, 300, 4, K=11008)) == "gemm_mx_a4w4_sq_kernel<64x64>"  # fp4 activations, K % 512 != 0 but K % 256 == 0: the 64 x 64 tiles take it (round 4)
    assert name(args(17, 4, 600, 4, K=11008)) == "mx_rows_a4w4_kernel<64x16>"   # ... above their row range: no tile kernel -> 64-row tiles of the few-row kernel
    assert name(args(17, 4, 300, 4, K=4096)) == "gemm_mx_a4w4_sq_kernel<64x64>"
    assert name(args(17, 4, 600, 4, K=4096)) == "gemm_mx_a4w4_sq_kernel<64x64>"   # (late round 6: fp4 weights up to 1024 rows while N K <= 4096^2)
    assert name(args(17, 4, 1100, 4, K=4096)) == "gemm_mx_a4w4_kernel<128x128>"
    assert name(args(16, 8, 5, 4)) == "mx_rows_a8w8_kernel<16x16>"       # round 4: 5 .. 64 rows, 16-column blocks
    assert name(args(16, 4, 33, 2)) == "mx_rows_a8w4_kernel<64x16>"
    assert name(args(17, 4, 20, 4)) == "mx_rows_a4w4_kernel<32x16>"
    assert name(args(16, 8, 64, 4, N=16384, K=16384)) == "gemm_mx_a8w8_sq_kernel<64x64>"  # past the x re-read budget: the tile kernel
    a = args(16, 8, 1, 4)
    a.tuning[0] = 2                                                       # A/B switch: MFMA kernel at decode sizes
    assert name(a) == "gemm_mx_a8w8_kernel<32x128>"
    assert name(args(17, 4, 48, 4, N=16384)) == "gemm_mx_a4w4_sq_kernel<64x64>"   # 48 x 2 KB x 1024 blocks of x re-reads: over the budget
    assert name(args(17, 4, 40, 4, N=16384)) == "gemm_mx_a4w4_sq_kernel<64x64>"
    # NVFP4: no scaled-MFMA form takes e4m3 block-16 scales; both operands are exact in fp16, so the fp16 tile kernel runs it (round 4:
    # x expanded by a kernel in front, the weights in the K loop; workspace = tickets + slabs + M K fp16 + M floats)
    assert name(args(18, 4, 8, 4, group=16)) == "nvfp4_rows_kernel<16x16>"   # 1 .. 64 rows: both operands expanded in registers
    assert name(args(18, 4, 40, 4, group=16)) == "gemm_nvfp4_f16_kernel<64x64>"
    assert name(args(18, 4, 256, 4, group=16)) == "gemm_nvfp4_f16_kernel<64x64>"
    a = args(18, 4, 256, 4, group=16)
    a.tuning[2] = 32                                                      # round 4, late: the narrow 64 x 64 tiles (forced)
    assert name(a) == "gemm_nvfp4_f16_kernel<64x64>"
    a = args(15, 4, 100, 0)
    a.tuning[2] = 32
    assert name(a) == "gemm_a16w4_mxfp_kernel<64x64>"
    a = args(15, 8, 100, 0)
    a.tuning[1], a.tuning[2] = 2, 32
    assert name(a) == "gemm_a16w8_mxfp_kernel<64x64>"
    a = args(18, 4, 8, 4, group=16)
    a.tuning[0] = 2                                                       # A/B switch: the tile kernel at any M
    assert name(a) == "gemm_nvfp4_f16_kernel<32x128>"
    assert lib.gemlite_hip_workspace_bytes(C.byref(a)) >= 65536 * 4 + 8 * 4096 * 2 + 256
    a.tuning[0] = 1
    assert name(a) == "mx_generic_kernel"                                 # A/B switch: the coverage kernel of rounds 2-3
    assert name(args(18, 4, 8, 4, group=16, N=4096 + 64)) == "nvfp4_rows_kernel<16x16>"   # the rows kernel only needs N % 16 == 0
    assert name(args(18, 4, 300, 4, group=16, N=4096 + 64)) == "mx_generic_kernel"        # above 64 rows with N % 128 != 0: coverage
    assert lib.gemlite_hip_query(C.byref(args(18, 4, 8, 4, group=32))) == _hip.ERR_UNSUPPORTED
    assert lib.gemlite_hip_query(C.byref(args(17, 8, 8, 4))) == _hip.ERR_UNSUPPORTED  # fp4 activations x fp8 weights
    a = args(16, 8, 64, 4)
    a.stride_wk, a.stride_wn = 4096, 1  # not K-contiguous: coverage kernel
    assert name(a) == "mx_generic_kernel"
