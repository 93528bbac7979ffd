"""Block-scaled formats on the GPU (`pytest -m gpu`): activation quantisers bit-exact against the oracle and the
reference's golden outputs; the MX matmul kernels against the float64 oracle on the very tensors the layer holds
and the activations its own quantiser produced; the reference's acceptance bar (tests/test_mxfp.py).

Tolerance of the matmul comparisons: the kernels multiply the SAME quantised operands as the oracle and accumulate in
fp32, so the only differences are the accumulation order and the rounding of the output to bf16 / fp16:
mean|err| / mean|y| < 4e-3 (bf16 out) / 1e-3 (fp16 out), elementwise |err| <= 10 tol mean|y| + 4 tol |y|.
"""
import os

import numpy as np
import pytest
import torch

import gemlite_amd
from gemlite_amd import DType, _hip, helper as H
from gemlite_amd import core as C
from gemlite_amd.quant_utils import (scale_activations_mxfp4, scale_activations_mxfp8, scale_activations_nvfp4,
                                     scale_activations_per_token)
from oracle import mx_oracle as MX
from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
Z = np.load(os.path.join(GOLDEN, "mx.npz"))
REL_TOL = {torch.float16: 1.0e-3, torch.bfloat16: 4.0e-3}


def _bytes(t):
    return t.contiguous().view(torch.uint8).cpu().numpy()


# ------------------------------------------------------------------------------------------------ activation quantisers
QUANT = {"mxfp8": scale_activations_mxfp8, "mxfp4": scale_activations_mxfp4, "nvfp4": scale_activations_nvfp4}


@pytest.mark.parametrize("tag", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["mxfp8", "mxfp4", "nvfp4"])
def test_activation_quantisers_match_the_reference_golden(tag, name):
    a = torch.from_numpy(np.ascontiguousarray(Z[f"act_{tag}_x"]))
    x = (a.view(torch.bfloat16) if tag == "bf16" else a).to(DEV)
    y, s = QUANT[name](x)
    assert np.array_equal(_bytes(s), Z[f"act_{tag}_{name}_s"]), "block scales differ from the reference"
    assert np.array_equal(_bytes(y), Z[f"act_{tag}_{name}_y"]), "elements differ from the reference"


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("name", ["mxfp8", "mxfp4", "nvfp4"])
@pytest.mark.parametrize("M,K", [(1, 4096), (37, 1024), (256, 2048)])
def test_activation_quantisers_bit_exact_vs_oracle(tdt, name, M, K):
    g = torch.Generator().manual_seed(M * 7 + K)
    x = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).to(tdt)
    x[0, :32] = 0
    if M > 1:
        x[1, 7] = 300.0
    y, s = QUANT[name](x.to(DEV))
    yo, so = getattr(MX, "scale_activations_" + name)(x.float().numpy())
    assert tuple(s.shape) == so.shape and tuple(y.shape) == yo.shape
    assert np.array_equal(_bytes(s), so)
    assert np.array_equal(_bytes(y), yo)


# ------------------------------------------------------------------------------------------------ matmul
def _kernel_name(layer, x, tuning=(0, 0, 0, 0)):
    from gemlite_amd.core import _static_args
    a = _static_args(layer.W_q, layer.scales, layer.zeros, layer.get_meta_args())
    a.matmul_type, a.M = -1, x.reshape(-1, x.shape[-1]).shape[0]
    a.x = a.out = a.scales_x = 0x1000
    code = layer.input_dtype.value
    K = layer.in_features
    a.stride_xm, a.stride_xk = (K // 2 if code in (17, 18) else K), 1
    a.stride_om, a.stride_on = a.N, 1
    a.stride_sx_m = K // layer.group_size
    a.input_dtype = code
    for i in range(4):
        a.tuning[i] = tuning[i]
    return _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()


def _weights_nk(layer):
    """element values [N, K] (float32) and scale bytes [N, K/g] of a packed block-scaled layer"""
    N, K = layer.out_features, layer.in_features
    wq = layer.W_q.data
    if wq.dtype == torch.uint8:
        vals = MX.fp4_unpack(wq.t().contiguous().cpu().numpy())          # [N, K/2] bytes -> [N, K]
    else:
        vals = MX.fp8_e4m3_decode(wq.t().contiguous().view(torch.uint8).cpu().numpy())
    sc = layer.scales.data.contiguous().view(torch.uint8).cpu().numpy()  # [N, K/g]
    assert vals.shape == (N, K) and sc.shape == (N, K // layer.group_size)
    return vals, sc


def _oracle(layer, x):
    """float64 result on the layer's own tensors; activations quantised by the package's (bit-exact) quantiser"""
    code, c_mode, g = layer.input_dtype, layer.channel_scale_mode, layer.group_size
    wv, ws = _weights_nk(layer)
    x2 = x.reshape(-1, x.shape[-1])
    nv = code == DType.NVFP4
    if code in (DType.MXFP16, DType.MXBF16):
        return MX.mx_matmul(x2.float().cpu().numpy(), wv, sw=ws, group=g)
    if code == DType.MXFP8 and c_mode == 2:
        xq, sx = scale_activations_per_token(x2, w_dtype=torch.float8_e4m3fn)
        return MX.mx_matmul(MX.fp8_e4m3_decode(_bytes(xq)), wv, sw=ws, group=g, scales_x_token=sx.cpu().numpy())
    if code == DType.MXFP8:
        xq, sx = scale_activations_mxfp8(x2)
        return MX.mx_matmul(MX.fp8_e4m3_decode(_bytes(xq)), wv, sx=_bytes(sx), sw=ws, group=g)
    xq, sx = (scale_activations_nvfp4 if nv else scale_activations_mxfp4)(x2)
    return MX.mx_matmul(MX.fp4_unpack(_bytes(xq)), wv, sx=_bytes(sx), sw=ws, group=g, e4m3_scales=nv,
                        post=0.05 ** 2 if nv else 1.0)


def _check(tag, y, ref, tdt, tol_scale=1.0):
    y = y.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64).reshape(y.shape)
    err = np.abs(y - ref)
    scale = max(float(np.abs(ref).mean()), 1e-12)
    tol = REL_TOL[tdt] * tol_scale
    assert np.isfinite(y).all(), tag
    assert err.mean() / scale < tol, (tag, err.mean() / scale)
    viol = err > 10 * tol * scale + 4 * tol * np.abs(ref)
    assert int(viol.sum()) == 0, (tag, int(viol.sum()), float(err.max()), scale,
                                  [int(v) for v in np.unravel_index(int(err.argmax()), err.shape)])


def _linear(N, K, tdt, seed):
    g = torch.Generator().manual_seed(seed)
    lin = torch.nn.Linear(K, N, bias=True, dtype=tdt)
    with torch.no_grad():
        lin.weight.copy_((torch.randn(N, K, generator=g) / 10).to(tdt))
        lin.weight[:, :32] *= 8          # blocks with very different scales
        lin.bias.copy_((torch.randn(N, generator=g) / 10).to(tdt))
    return lin.to(DEV)


PROCS = {
    "A16W8_MXFP": lambda tdt: H.A16W8_MXFP(device=DEV, dtype=tdt),
    "A16W4_MXFP": lambda tdt: H.A16W4_MXFP(device=DEV, dtype=tdt),
    "A8W8_MXFP_dynamic_post": lambda tdt: H.A8W8_MXFP_dynamic(device=DEV, dtype=tdt, post_scale=True),
    "A8W8_MXFP_dynamic": lambda tdt: H.A8W8_MXFP_dynamic(device=DEV, dtype=tdt, post_scale=False),
    "A8W4_MXFP_dynamic_post": lambda tdt: H.A8W4_MXFP_dynamic(device=DEV, dtype=tdt, post_scale=True),
    "A8W4_MXFP_dynamic": lambda tdt: H.A8W4_MXFP_dynamic(device=DEV, dtype=tdt, post_scale=False),
    "A4W4_MXFP_dynamic": lambda tdt: H.A4W4_MXFP_dynamic(device=DEV, dtype=tdt),
    "A4W4_NVFP_dynamic": lambda tdt: H.A4W4_NVFP_dynamic(device=DEV, dtype=tdt),
}
EXPECT = {"A16W8_MXFP": "gemm_a16w8_mxfp_kernel", "A16W4_MXFP": "gemm_a16w4_mxfp_kernel",
          "A8W8_MXFP_dynamic": "gemm_mx_a8w8_kernel", "A8W8_MXFP_dynamic_post": "gemm_mx_a8w8_kernel",
          "A8W4_MXFP_dynamic": "gemm_mx_a8w4_kernel", "A8W4_MXFP_dynamic_post": "gemm_mx_a8w4_kernel",
          "A4W4_MXFP_dynamic": "gemm_mx_a4w4_kernel", "A4W4_NVFP_dynamic": "gemm_nvfp4_f16_kernel"}


@pytest.mark.parametrize("proc", list(PROCS))
@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
def test_processor_forward_vs_oracle(proc, tdt):
    """every MXFP / NVFP processor, N = 512, K = 1024, M from decode to a ragged prefill tile, with bias"""
    N, K = 512, 1024
    lin = _linear(N, K, tdt, seed=len(proc))
    bias = lin.bias.data.clone()
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    g = torch.Generator().manual_seed(11)
    for M in (1, 3, 16, 33, 100, 256):
        x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
        name = _kernel_name(layer, x)
        want = EXPECT[proc] if (M > 4 or "NVFP" in proc) else "mx_gemv_w"
        if M <= 64 and "dynamic" in proc and "NVFP" not in proc:
            want = "mx_rows_"  # round 4: 1 .. 64 rows of the fp8 / fp4 activation formats
        if M <= 64 and "NVFP" in proc:
            want = "nvfp4_rows_"  # round 4: both operands expanded to fp16 in registers, two v_mfma_f32_16x16x32_f16 per chunk and 16 rows
        if (M > 64 or (M > 22 and proc == "A4W4_MXFP_dynamic")) and "dynamic" in proc and "NVFP" not in proc:
            want = EXPECT[proc].replace("_kernel", "_sq_kernel")  # round 4: 65 .. 384 rows on 64 x 64 tiles, K unsplit
        if M <= 64 and proc.startswith("A16"):
            want = "a16w8_mxfp_rows_kernel" if "W8" in proc else "a16w4_mxfp_rows_kernel"  # round 4: the weight-only layers on the A16W8 rows kernel
        assert name.startswith(want), (proc, M, name)
        y = layer(x)
        assert y.dtype == tdt and tuple(y.shape) == (M, N)
        ref = _oracle(layer, x) + bias.float().cpu().numpy().astype(np.float64)
        _check(f"{proc} {tdt} M={M} {name}", y, ref, tdt)
        if M <= 4 and "dynamic" in proc and "NVFP" not in proc:  # the streaming kernel of rounds 2-3 (A/B switch) gives the same answer
            try:
                C.TUNING_OVERRIDE = (5, 0, 0, 0)
                assert _kernel_name(layer, x, (5, 0, 0, 0)).startswith("mx_gemv_w")
                _check(f"{proc} {tdt} M={M} gemv", layer(x), ref, tdt)
            finally:
                C.TUNING_OVERRIDE = None
        if M <= 4 and proc.startswith("A16"):  # the streaming kernel of rounds 2-3 (A/B switch) gives the same answer
            try:
                C.TUNING_OVERRIDE = (5, 0, 0, 0)
                assert _kernel_name(layer, x, (5, 0, 0, 0)).startswith("mx_gemv_w")
                _check(f"{proc} {tdt} M={M} gemv", layer(x), ref, tdt)
            finally:
                C.TUNING_OVERRIDE = None
        if M <= 4 and "NVFP" not in proc:  # the MFMA kernel at decode sizes (A/B switch) gives the same answer
            try:
                C.TUNING_OVERRIDE = (2, 0, 0, 0)
                assert _kernel_name(layer, x, (2, 0, 0, 0)).startswith(EXPECT[proc])
                _check(f"{proc} {tdt} M={M} mfma", layer(x), ref, tdt)
            finally:
                C.TUNING_OVERRIDE = None


@pytest.mark.parametrize("proc", ["A8W8_MXFP_dynamic", "A8W4_MXFP_dynamic", "A4W4_MXFP_dynamic", "A8W8_MXFP_dynamic_post",
                                  "A16W8_MXFP", "A16W4_MXFP", "A4W4_NVFP_dynamic"])
def test_mfma_kernel_tiles_slices_and_coverage_agree(proc):
    """the scaled-MFMA kernel at every tile height x K slices, and the coverage kernel, against the oracle and each other"""
    tdt = torch.bfloat16
    N, K = 256, 4096
    lin = _linear(N, K, tdt, seed=3)
    lin.bias = None
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(150, K, generator=g) / 4).to(tdt).to(DEV)
    ref = _oracle(layer, x)
    outs = {}
    try:
        tunings = [(0, 1, 1, 0), (0, 1, 2, 0), (0, 1, 4, 0), (0, 2, 1, 0), (0, 4, 2, 0), (0, 3, 4, 0), (0, 8, 1, 0), (1, 0, 0, 0)]
        if proc.startswith("A16") or "NVFP" in proc:
            tunings += [(0, 1, 8, 0), (0, 5, 8, 0)]  # 256-row tiles exist on the 16-bit-activation kernel only (NVFP4 runs on it: both operands expanded to fp16)
        if proc.startswith("A16") or "NVFP" in proc:
            tunings += [(0, 1, 32, 0), (0, 3, 32, 0)]  # round 4, late: the narrow 64 x 64 tiles (KH = 4) of the same template, unsplit and with K slices
        for tuning in tunings:
            C.TUNING_OVERRIDE = tuning
            name = _kernel_name(layer, x, tuning)
            assert name.startswith("mx_generic_kernel" if tuning[0] == 1 else EXPECT[proc]), (tuning, name)
            assert ("<64x64>" in name) == (tuning[2] == 32), (tuning, name)
            y = layer(x)
            _check(f"{proc} tuning={tuning} {name}", y, ref, tdt)
            outs[tuning] = y.float().cpu()
            y2 = layer(x)  # run-to-run deterministic (fixed slice order in the combine)
            assert torch.equal(y2.float().cpu(), outs[tuning]), tuning
    finally:
        C.TUNING_OVERRIDE = None
    # split-K counters are left at zero
    for ws in _hip._workspaces.values():
        torch.cuda.synchronize()
        assert int(ws[:4 * 61440].view(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize("proc", ["A8W8_MXFP_dynamic", "A8W8_MXFP_dynamic_post", "A8W4_MXFP_dynamic", "A8W4_MXFP_dynamic_post", "A4W4_MXFP_dynamic"])
def test_few_row_scaled_mfma_kernel(proc):
    """mx_rows_kernel (one v_mfma_scale_f32_16x16x128_f8f6f4 per 128-k chunk and 16 rows, operands straight from memory): every
    format pair (fp8 x fp8, fp8 x fp4 — each side in its own operand layout — fp4 x fp4), block and per-token activation scales, the
    three row-tile heights with ragged M, K an odd multiple of 128, forced at decode sizes; against the float64 oracle and against the
    8-wave tile kernel it replaces at these sizes."""
    tdt = torch.bfloat16
    N, K = 256, 1152
    lin = _linear(N, K, tdt, seed=9)
    bias = lin.bias.data.float().cpu().numpy().astype(np.float64)
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    g = torch.Generator().manual_seed(13)
    for M in (1, 3, 5, 16, 17, 32, 33, 50, 64):
        x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
        x[0, :64] *= 30   # blocks with very different scales inside a row
        ref = _oracle(layer, x) + bias
        try:
            C.TUNING_OVERRIDE = (4, 0, 0, 0)
            name = _kernel_name(layer, x, (4, 0, 0, 0))
            assert name.startswith("mx_rows_") and name.endswith({1: "<16x16>", 2: "<32x16>", 4: "<64x16>"}[1 if M <= 16 else (2 if M <= 32 else 4)]), name
            y = layer(x)
            _check(f"{proc} rows M={M} {name}", y, ref, tdt)
            C.TUNING_OVERRIDE = (2, 0, 0, 0)
            y_tile = layer(x)
            _check(f"{proc} rows vs tile kernel M={M}", y, y_tile.float().cpu().numpy(), tdt)
        finally:
            C.TUNING_OVERRIDE = None
        assert _kernel_name(layer, x).startswith("mx_rows_"), _kernel_name(layer, x)


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("proc", ["A8W8_MXFP_dynamic", "A8W4_MXFP_dynamic", "A4W4_MXFP_dynamic", "A8W8_MXFP_dynamic_post", "A8W4_MXFP_dynamic_post", "A4W4_NVFP_dynamic"])
def test_one_row_is_quantised_inside_the_few_row_kernel(proc, tdt):
    """M = 1 of the block-scaled dynamic layers: `layer(x)` is ONE launch — mx_rows_kernel<..., FQ> requests its weights, quantises the row
    block by block into LDS (mx_quant_block = the arithmetic of the quantiser kernel; `_post`: one fp32 scale per token, the processors'
    default) and multiplies.  Bit-identical to quantiser + the same kernel on the quantised row; 2-d / 3-d / 1-d inputs, rows with very
    different block magnitudes, K = 1152 and 4096; the switch."""
    for N, K in ((256, 1152), (512, 4096)):
        lin = _linear(N, K, tdt, seed=31)
        lin.bias = None
        layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
        g = torch.Generator().manual_seed(19)
        for rep, shape in enumerate(((1, K), (1, 1, K), (K,))):
            x = (torch.randn(*shape, generator=g) * (0.1 + 0.3 * rep)).to(tdt).to(DEV)
            x.view(-1)[:32] *= 40.0
            x.view(-1)[64:96] = 0.0  # an all-zero block: the floored scale
            a = C._static_args(layer.W_q, layer.scales, layer.zeros, layer.get_meta_args())
            a.matmul_type, a.M, a.x, a.out = -1, 1, x.data_ptr(), 0x1000
            a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = K, 1, N, 1
            a.input_dtype = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
            name = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
            assert name.startswith(("mx_rows_", "nvfp4_rows_")) and "fused_quant" in name, name
            y_fused = layer(x)
            C.FUSE_ACT_QUANT_M1 = False
            try:
                y_two = layer(x)
            finally:
                C.FUSE_ACT_QUANT_M1 = True
            torch.cuda.synchronize()
            assert y_fused.shape == y_two.shape and torch.equal(y_fused, y_two), (proc, tdt, N, K, shape, float((y_fused.float() - y_two.float()).abs().max()))
        _check(f"{proc} fused M=1 {name}", layer(x).reshape(1, N), _oracle(layer, x.reshape(1, K)), tdt)


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("proc", ["A16W8_MXFP", "A16W4_MXFP"])
def test_weight_only_block_scaled_layers_on_the_rows_kernel(proc, tdt):
    """A16W8_MXFP / A16W4_MXFP (helper.py:372-400) for 1 .. 64 rows on a16w8_rows_kernel<MXW8 / MXW4> (round 4): fp8 / fp4 weights turned
    into the activation type by the scaled converters (the e8m0 block scale applied exactly), two v_mfma_f32_16x16x32 per 64-k chunk;
    every row-tile height with ragged M, K an odd multiple of 64, against the float64 oracle and against the tile kernel it replaces."""
    N, K = 256, 1088
    lin = _linear(N, K, tdt, seed=27)
    bias = lin.bias.data.float().cpu().numpy().astype(np.float64)
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    g = torch.Generator().manual_seed(23)
    for M in (1, 3, 16, 17, 32, 50, 64):
        x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
        ref = _oracle(layer, x) + bias
        name = _kernel_name(layer, x)
        assert name.startswith("a16w8_mxfp_rows_kernel" if "W8" in proc else "a16w4_mxfp_rows_kernel") and \
            name.endswith("<16x16>" if M <= 16 else ("<32x16>" if M <= 32 else "<64x16>")), (M, name)
        y = layer(x)
        _check(f"{proc} rows M={M} {name}", y, ref, tdt)
    x = (torch.randn(40, 1152, generator=g) / 4).to(tdt).to(DEV)  # K % 128 == 0: the tile kernel applies (A/B switch)
    lin2 = _linear(N, 1152, tdt, seed=28)
    lin2.bias = None
    layer2 = PROCS[proc](tdt).from_linear(lin2, del_orig=False)
    y_rows = layer2(x)
    try:
        C.TUNING_OVERRIDE = (2, 0, 0, 0)
        assert _kernel_name(layer2, x, (2, 0, 0, 0)).startswith("gemm_a16w")
        _check(f"{proc} rows vs tile kernel", y_rows, layer2(x).float().cpu().numpy(), tdt)
    finally:
        C.TUNING_OVERRIDE = None


def test_nvfp4_few_row_kernel_against_the_oracle_and_the_tile_kernel():
    """nvfp4_rows_kernel: every row-tile height with ragged M, K an odd multiple of 64 (no tile kernel there: K % 128 != 0), bf16 and fp16
    layers; against the float64 oracle, and where both apply against gemm_nvfp4_f16_kernel (tuning[0] = 2)."""
    for tdt in (torch.bfloat16, torch.float16):
        for N, K in ((256, 1088), (256, 1152)):
            lin = _linear(N, K, tdt, seed=33)
            bias = lin.bias.data.float().cpu().numpy().astype(np.float64)
            layer = PROCS["A4W4_NVFP_dynamic"](tdt).from_linear(lin, del_orig=False)
            g = torch.Generator().manual_seed(29)
            for M in (2, 16, 17, 32, 50, 64):
                x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
                name = _kernel_name(layer, x)
                assert name == "nvfp4_rows_kernel<%s>" % ("16x16" if M <= 16 else ("32x16" if M <= 32 else "64x16")), (M, name)
                y = layer(x)
                _check(f"nvfp4 rows {tdt} K={K} M={M}", y, _oracle(layer, x) + bias, tdt)
                if K % 128 == 0:
                    try:
                        C.TUNING_OVERRIDE = (2, 0, 0, 0)
                        assert _kernel_name(layer, x, (2, 0, 0, 0)).startswith("gemm_nvfp4_f16_kernel")
                        _check(f"nvfp4 rows vs tile M={M}", y, layer(x).float().cpu().numpy(), tdt)
                    finally:
                        C.TUNING_OVERRIDE = None


def test_fp4_activations_with_k_not_a_multiple_of_512_leave_the_coverage_kernel():
    """The fp4 x fp4 tile kernels step 512 k; K = 11008 (Llama down_proj) is 21.5 such steps and ran on the coverage kernel in rounds 2-3
    (4.5 ms at 4096 x 11008).  Round 4: any M on 64-row tiles of the few-row kernel (grid.y), 128-k chunks; 65 .. 512 rows on the 64 x 64
    tile kernel (256-k steps: K = 1280 and 11008 are whole numbers of them; late round 6: up to 1024 rows), K % 256 != 0 still on the few-row kernel."""
    tdt = torch.bfloat16
    N, K = 256, 1280
    lin = _linear(N, K, tdt, seed=21)
    bias = lin.bias.data.float().cpu().numpy().astype(np.float64)
    layer = PROCS["A4W4_MXFP_dynamic"](tdt).from_linear(lin, del_orig=False)
    g = torch.Generator().manual_seed(17)
    for M in (7, 64, 65, 100, 300, 600):
        x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
        name = _kernel_name(layer, x)
        assert name.startswith("gemm_mx_a4w4_sq_kernel" if 22 < M <= 1024 else "mx_rows_a4w4_kernel"), name   # (late round 6: up to 1024 rows while N K <= 4096^2)
        _check(f"a4w4 K=1280 M={M} {name}", layer(x), _oracle(layer, x) + bias, tdt)
    lin2 = _linear(N, K + 128, tdt, seed=22)
    layer2 = PROCS["A4W4_MXFP_dynamic"](tdt).from_linear(lin2, del_orig=False)
    x = (torch.randn(100, K + 128, generator=g) / 4).to(tdt).to(DEV)
    assert _kernel_name(layer2, x).startswith("mx_rows_a4w4_kernel"), _kernel_name(layer2, x)
    _check("a4w4 K=1408 M=100", layer2(x), _oracle(layer2, x) + lin2.bias.data.float().cpu().numpy().astype(np.float64), tdt)


@pytest.mark.parametrize("M", [1, 4, 16])
@pytest.mark.parametrize("proc,tol", [("A16W8_MXFP", 2e-4), ("A8W8_MXFP_dynamic", 2e-4), ("A16W4_MXFP", 7e-4),
                                      ("A8W4_MXFP_dynamic", 7e-4), ("A4W4_MXFP_dynamic", 1e-3), ("A4W4_NVFP_dynamic", 1e-3)])
def test_reference_acceptance_bar(proc, tol, M):
    """tests/test_mxfp.py of the reference: a 4096 -> 2048 bf16 linear (weights / 10), x = randn / 10, both manual kernel
    families, mean |y - linear(x)| below the reference's own tolerance."""
    tdt = torch.bfloat16
    torch.manual_seed(0)
    lin = torch.nn.Linear(4096, 2048, bias=False, device=DEV, dtype=tdt)
    lin.weight.data /= 10.0
    lin.weight.requires_grad = False
    torch.manual_seed(0)
    x = torch.randn((M, 4096), dtype=tdt, device=DEV) / 10.0
    kw = dict(post_scale=False) if proc == "A8W8_MXFP_dynamic" else {}
    layer = getattr(H, proc)(device=DEV, dtype=tdt, **kw).from_linear(lin, del_orig=False)
    y_ref = lin(x)
    for mt in ("GEMM_SPLITK", "GEMM"):
        y = layer.forward_manual(x, matmul_type=mt)
        err = (y_ref - y).abs().mean().item()
        assert err < tol, (proc, M, mt, err)
    assert layer.W_q.numel() * layer.W_q.element_size() == 4096 * 2048 // (1 if "W8" in proc else 2)


@pytest.mark.parametrize("proc,kname", [("A8W8_MXFP_dynamic", "gemm_mx_a8w8_tile_kernel"), ("A4W4_MXFP_dynamic", "gemm_mx_a4w4_tile_kernel"),
                                        ("A8W8_MXFP_dynamic_post", "gemm_mx_a8w8_tile_kernel"), ("A8W4_MXFP_dynamic", "gemm_mx_a8w4_tile_kernel"),
                                        ("A8W4_MXFP_dynamic_post", "gemm_mx_a8w4_tile_kernel")])
@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
def test_prefill_tile_kernel_vs_oracle(proc, kname, tdt):
    """256 x 256 tiles with both operands through LDS, forced (tuning[0] = 3) on ragged M (the planner picks it from ~100 tiles:
    test_prefill_tile_kernel_is_the_default_when_the_tiles_fill_the_chip)"""
    N, K = 512, 2048
    lin = _linear(N, K, tdt, seed=21)
    bias = lin.bias.data.clone()
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    g = torch.Generator().manual_seed(13)
    for M, tuning in ((700, (3, 0, 0, 0)), (512, (3, 0, 0, 0)), (300, (3, 0, 0, 0)), (33, (3, 0, 0, 0))):
        x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
        try:
            C.TUNING_OVERRIDE = tuning if any(tuning) else None
            name = _kernel_name(layer, x, tuning)
            assert name.startswith(kname), (M, tuning, name)
            y = layer(x)
        finally:
            C.TUNING_OVERRIDE = None
        ref = _oracle(layer, x) + bias.float().cpu().numpy().astype(np.float64)
        _check(f"{proc} {tdt} M={M} {name}", y, ref, tdt)
        try:  # and the 128-row kernel gives the same answer
            C.TUNING_OVERRIDE = (2, 0, 0, 0)
            _check(f"{proc} {tdt} M={M} 8-wave kernel", layer(x), ref, tdt)
        finally:
            C.TUNING_OVERRIDE = None


@pytest.mark.parametrize("proc,kname", [("A8W8_MXFP_dynamic", "gemm_mx_a8w8_sq_kernel"), ("A4W4_MXFP_dynamic", "gemm_mx_a4w4_sq_kernel"),
                                        ("A8W8_MXFP_dynamic_post", "gemm_mx_a8w8_sq_kernel"), ("A8W4_MXFP_dynamic", "gemm_mx_a8w4_sq_kernel"),
                                        ("A8W4_MXFP_dynamic_post", "gemm_mx_a8w4_sq_kernel")])
@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
def test_unsplit_64x64_tile_kernel_vs_oracle(proc, kname, tdt):
    """gemm_mx_sq_kernel (round 4): 64 x 64 tiles, K unsplit, both operands and the block scales through LDS — the default between 65 and
    ~256 rows where its tiles fill the chip about once; forced (tuning[0] = 6) on ragged M at every stage depth, K = 256 (one step,
    shorter than the pipeline) and a K that is an odd number of steps; against the oracle and the 128-column kernel (tuning[0] = 2)."""
    g = torch.Generator().manual_seed(17)
    for N, K, Ms in ((512, 2048 + 256, (300, 65, 33)), (192, 256, (100,)), (1024, 1024, (256,))):
        lin = _linear(N, K, tdt, seed=23)
        bias = lin.bias.data.clone()
        layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
        for M in Ms:
            x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
            ref = _oracle(layer, x) + bias.float().cpu().numpy().astype(np.float64)
            if M > 64:
                assert _kernel_name(layer, x).startswith(kname), (M, _kernel_name(layer, x))  # the default here
            for tuning in ((6, 0, 0, 0), (6, 0, 2, 0), (6, 0, 3, 0), (6, 0, 4, 0)):
                try:
                    C.TUNING_OVERRIDE = tuning
                    name = _kernel_name(layer, x, tuning)
                    assert name.startswith(kname), (M, tuning, name)
                    y = layer(x)
                finally:
                    C.TUNING_OVERRIDE = None
                _check(f"{proc} {tdt} {N}x{K} M={M} {tuning} {name}", y, ref, tdt)
            if K % (512 if "A4" in proc else 256) == 0:
                try:  # and the 128-column kernel gives the same answer
                    C.TUNING_OVERRIDE = (2, 0, 0, 0)
                    assert "sq_kernel" not in _kernel_name(layer, x, (2, 0, 0, 0))
                    _check(f"{proc} {tdt} M={M} 8-wave kernel", layer(x), ref, tdt)
                finally:
                    C.TUNING_OVERRIDE = None


def test_prefill_tile_kernel_is_the_default_when_the_tiles_fill_the_chip():
    tdt = torch.bfloat16
    N, K, M = 4096, 512, 1536   # 16 x 6 = 96 tiles of 256 x 256
    lin = _linear(N, K, tdt, seed=31)
    lin.bias = None
    layer = H.A4W4_MXFP_dynamic(device=DEV, dtype=tdt).from_linear(lin, del_orig=False)
    x = (torch.randn(M, K, generator=torch.Generator().manual_seed(3)) / 4).to(tdt).to(DEV)
    assert _kernel_name(layer, x).startswith("gemm_mx_a4w4_tile_kernel")
    _check("tile kernel, natural", layer(x), _oracle(layer, x), tdt)
    assert _kernel_name(layer, x[:1100]).startswith("gemm_mx_a4w4_kernel")  # 80 tiles: the 128-row kernel (up to 1024 rows: the unsplit 64 x 64 tiles, late round 6)
    assert _kernel_name(layer, x[:1024]).startswith("gemm_mx_a4w4_sq_kernel")


@pytest.mark.parametrize("proc", ["A4W4_MXFP_dynamic", "A8W8_MXFP_dynamic_post", "A16W4_MXFP", "A4W4_NVFP_dynamic"])
def test_autotune_layer_on_block_scaled_layers(proc):
    """helper.autotune_layer() on block-scaled layers (round 4, late): the candidates of include/gemlite_hip.h's block-scaled row are timed,
    the winner is filed under the `mx` family lookup_tuning() reads, and the tuned layer still matches the oracle.  One row of a dynamic
    layer is not tuned (it is quantised inside the few-row kernel)."""
    tdt = torch.float16 if "NVFP" in proc else torch.bfloat16
    N, K = 1024, 2048
    lin = _linear(N, K, tdt, seed=41)
    lin.bias = None
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    C.GemLiteLinear.reset_config()
    try:
        res = H.autotune_layer(layer, batch_sizes=(1, 8, 200), iters=5)
        assert set(res) == ({1, 8, 200} if proc.startswith("A16") else {8, 200}), res.keys()
        assert all(len(v["tuning"]) == 4 and v["us"] > 0 and len(v["candidates"]) >= 3 for v in res.values())
        g = torch.Generator().manual_seed(9)
        for M in res:
            x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
            a = C._static_args(layer.W_q, layer.scales, layer.zeros, layer.get_meta_args())
            assert C.lookup_tuning(-1, M, a) == tuple(res[M]["tuning"]), (M, res[M])
            _check(f"{proc} autotuned M={M} {res[M]['tuning']}", layer(x), _oracle(layer, x), tdt)
    finally:
        C.GemLiteLinear.reset_config()


def test_mx_layer_state_dict_round_trip_and_functional_op():
    tdt = torch.bfloat16
    lin = _linear(256, 512, tdt, seed=9)
    layer = H.A8W4_MXFP_dynamic(device=DEV, dtype=tdt, post_scale=False).from_linear(lin, del_orig=False)
    x = (torch.randn(5, 512) / 4).to(tdt).to(DEV)
    y = layer(x)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    fresh = gemlite_amd.GemLiteLinear(4, 32, 512, 256, DType.MXFP8, DType.BF16, scaled_activations=True)
    fresh.load_state_dict(sd)
    # `metadata` is written by pack(), BEFORE the processor sets the dynamic modes (a quirk shared with the reference,
    # helper.py:703-705 vs core.py:503-507): a loader re-applies them like the processor does
    assert fresh.get_meta_args()[:9] == layer.get_meta_args()[:9]
    fresh.W_group_mode, fresh.channel_scale_mode = 0, 4
    assert fresh.get_meta_args() == layer.get_meta_args()
    assert torch.equal(fresh(x), y)
    y_op = torch.ops.gemlite.forward_functional(x, layer.bias, layer.get_tensor_args(), layer.get_meta_args(), -1)
    assert torch.equal(y_op, y)


def test_large_shape_and_linearity():
    """A8W8 MXFP at 8192 x 8192, M = 256 (full-size property): agrees with the coverage kernel, and scaling x by 2 (exact
    in every format) scales the output by 2 bit for bit."""
    tdt = torch.bfloat16
    torch.manual_seed(1)
    lin = torch.nn.Linear(8192, 8192, bias=False, device=DEV, dtype=tdt)
    lin.weight.data /= 10.0
    layer = H.A8W8_MXFP_dynamic(device=DEV, dtype=tdt, post_scale=False).from_linear(lin, del_orig=True)
    x = (torch.randn(256, 8192, device=DEV) / 4).to(tdt)
    y = layer(x)
    assert torch.equal(layer(x * 2), y * 2)
    try:
        C.TUNING_OVERRIDE = (1, 0, 0, 0)
        y_cov = layer(x[:8])
    finally:
        C.TUNING_OVERRIDE = None
    _check("8192^2 vs coverage kernel", y[:8], y_cov.float().cpu().numpy(), tdt)


@pytest.mark.parametrize("proc", ["A16W4_MXFP", "A8W8_MXFP_dynamic", "A4W4_MXFP_dynamic", "A4W4_NVFP_dynamic"])
def test_odd_shapes_run_on_the_coverage_kernel(proc):
    """N not a multiple of 128, K a bare multiple of 32 (no MFMA tile divides them): correct on the coverage kernel, at
    decode and at batch sizes"""
    tdt = torch.bfloat16
    N, K = 192, 160
    lin = _linear(N, K, tdt, seed=4)
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    bias = lin.bias.data.float().cpu().numpy().astype(np.float64)
    g = torch.Generator().manual_seed(2)
    for M in (1, 7, 40):
        x = (torch.randn(M, K, generator=g) / 4).to(tdt).to(DEV)
        name = _kernel_name(layer, x)
        assert name in ("mx_generic_kernel", "mx_gemv_w4_kernel", "mx_gemv_w8_kernel"), name
        if M > 4:
            assert name == "mx_generic_kernel", name
        _check(f"{proc} odd M={M} {name}", layer(x), _oracle(layer, x) + bias, tdt)


def test_n_contiguous_mx_weights_take_the_coverage_kernel():
    """pack(contiguous=True) stores the fp4 bytes [K/2, N] N-contiguous: not the layout the MFMA kernels read, still correct"""
    tdt = torch.bfloat16
    N, K = 256, 512
    lin = _linear(N, K, tdt, seed=6)
    from gemlite_amd.quant_utils import WeightQuantizerMXFP
    wq, sc = WeightQuantizerMXFP(compute_dtype=tdt, device=DEV).quantize_mxfp4(lin.weight.data, index=True)
    layer = gemlite_amd.GemLiteLinear(4, 32, K, N, DType.MXFP4, DType.BF16, scaled_activations=True)
    layer.pack(wq.view(N, K), sc.view(N, K // 32), None, None, contiguous=True)
    layer.W_group_mode, layer.channel_scale_mode = 0, 4
    assert layer.W_q.is_contiguous() and layer.data_contiguous
    x = (torch.randn(9, K) / 4).to(tdt).to(DEV)
    assert _kernel_name(layer, x) == "mx_generic_kernel"
    _check("contiguous fp4", layer(x), _oracle(layer, x), tdt)


def test_mx_layer_under_torch_compile_and_graph_capture():
    tdt = torch.bfloat16
    lin = _linear(256, 512, tdt, seed=8)
    layer = H.A8W8_MXFP_dynamic(device=DEV, dtype=tdt, post_scale=False).from_linear(lin, del_orig=False)
    x = (torch.randn(16, 512) / 4).to(tdt).to(DEV)
    y = layer(x)

    class Mod(torch.nn.Module):
        def __init__(self, l):
            super().__init__()
            self.l = l

        def forward(self, t):
            return self.l(t) * 2

    yc = torch.compile(Mod(layer), fullgraph=True)(x)
    assert torch.equal(yc, y * 2)
    # hipGraph capture: the quantiser and the matmul are plain launches on the capture stream
    sstream = torch.cuda.Stream()
    sstream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sstream):
        layer(x)
    torch.cuda.current_stream().wait_stream(sstream)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=sstream):
        yg = layer(x)
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y)


def _random_mx_cases(n, seed):
    rng = np.random.default_rng(seed)
    procs = [p for p in PROCS]
    out = []
    for i in range(n):
        proc = procs[int(rng.integers(len(procs)))]
        N = int(rng.choice([64, 128, 192, 256, 384, 512, 1024]))
        K = int(rng.choice([32, 64, 96, 128, 256, 512, 640, 1024, 1536, 2048, 4096]))
        M = int(rng.choice([1, 2, 3, 4, 5, 8, 17, 32, 33, 64, 100, 129, 255, 300, 513, 600]))
        tdt = torch.float16 if rng.random() < 0.4 else torch.bfloat16
        out.append((i, proc, N, K, M, tdt))
    return out


@pytest.mark.parametrize("case", _random_mx_cases(64, seed=77), ids=lambda c: f"r{c[0]}-{c[1]}-{c[2]}x{c[3]}-M{c[4]}-{str(c[5])[6:]}")
def test_random_block_scaled_cases_against_the_oracle(case):
    """Seeded random sweep over processor x (N, K) (incl. sizes no MFMA tile divides) x M (every kernel family: decode, 8-wave
    scaled MFMA, 256 x 256 prefill tiles, 16-bit-activation MFMA, coverage) x output dtype, each against the float64 oracle."""
    i, proc, N, K, M, tdt = case
    if proc == "A4W4_NVFP_dynamic" and K % 32 != 0:
        pytest.skip("quantiser pieces are 32 k")
    lin = _linear(N, K, tdt, seed=100 + i)
    bias = lin.bias.data.float().cpu().numpy().astype(np.float64)
    layer = PROCS[proc](tdt).from_linear(lin, del_orig=False)
    g = torch.Generator().manual_seed(1000 + i)
    x = (torch.randn(M, K, generator=g) * (0.05 + 2 * float(torch.rand(1, generator=g)))).to(tdt).to(DEV)
    name = _kernel_name(layer, x)
    y = layer(x)
    _check(f"random mx {case} {name}", y, _oracle(layer, x) + bias, tdt)
