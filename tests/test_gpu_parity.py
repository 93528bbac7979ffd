"""GPU parity tests (run on a real MI355X via `pytest -m gpu`).  Every call goes through the product path:
gemlite_amd.GemLiteLinear -> ctypes -> libgemlite_hip.so (C ABI) -> HIP kernel.  The checker is the CPU
oracle (oracle/gemlite_oracle.py, float64) and the committed golden outputs of the reference's own kernels.

Tolerances (stated once, used everywhere):
  * REL_MEAN: mean|y - y_oracle| / mean|y_oracle|  <  1.0e-3 (fp16 out), 4.0e-3 (bf16 out), 1e-5 (fp32 out
    from integer accumulation), and the reference's own absolute gate mean|err| < 1e-3 (5e-3 for fp8)
    (tests/test_gemlitelineartriton.py:137,373) on its `gen_data`-style inputs;
  * integer / byte work (bit packing, int8 activation quant, int8 x int8 accumulation): bit-exact.
A JSON report of every comparison is written to gpurun_out/parity_report.json for offline reading.
"""
import json
import os

import numpy as np
import pytest
import torch

import gemlite_amd
from gemlite_amd import DType, GemLiteLinear, _hip, bitpack
from gemlite_amd.quant_utils import scale_activations_per_token
from oracle import gemlite_oracle as O
from tests.golden_util import as_torch, load_cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = []
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL_TOL = {1: 1.0e-3, 2: 4.0e-3, 0: 2.0e-4}  # by output dtype code


@pytest.fixture(scope="module", autouse=True)
def _report():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def _kernel_name(lin, x, mt=-1, tuning=(0, 0, 0, 0)):
    """Which kernel the C ABI picks (asks the library, launches nothing)."""
    from gemlite_amd.core import _static_args
    a = _static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
    a.matmul_type, a.M = mt, x.reshape(-1, x.shape[-1]).shape[0]
    a.x = a.out = 0x1000
    a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = x.shape[-1], 1, a.N, 1
    if lin.channel_scale_mode in (2, 3):
        a.scales_x = 0x1000
    a.input_dtype = lin.input_dtype.value
    for i in range(4):
        a.tuning[i] = tuning[i]
    return _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()


def _compare(tag, y, y_ref, out_code, abs_gate=1e-3, rel_tol=None, extra=None):
    y = y.detach().float().cpu().numpy().astype(np.float64)
    y_ref = np.asarray(y_ref, np.float64).reshape(y.shape)
    err = np.abs(y - y_ref)
    scale = max(float(np.abs(y_ref).mean()), 1e-12)
    rec = dict(tag=tag, mean_abs_err=float(err.mean()), max_abs_err=float(err.max()), mean_abs_ref=scale,
               rel_mean=float(err.mean() / scale), rel_max=float(err.max() / scale), finite=bool(np.isfinite(y).all()),
               argmax=[int(v) for v in np.unravel_index(int(err.argmax()), err.shape)])
    if extra:
        rec.update(extra)
    REPORT.append(rec)
    rel_tol = REL_TOL[out_code] if rel_tol is None else rel_tol
    # elementwise gate |err| <= atol + rtol * |y_ref|: catches a single wrong output that the mean gates would absorb.
    # atol = 10 rel_tol * mean|y_ref| is ~8 sigma of the dequantised-weight rounding noise for bf16 (more for fp16).
    atol, rtol = 10 * rel_tol * scale, 4 * rel_tol
    viol = err > atol + rtol * np.abs(y_ref)
    rec["elementwise_violations"] = int(viol.sum())
    assert rec["finite"], rec
    assert rec["rel_mean"] < rel_tol, rec
    assert rec["rel_max"] < 60 * rel_tol, rec
    assert rec["elementwise_violations"] == 0, rec
    if abs_gate is not None and scale < 5.0:
        assert rec["mean_abs_err"] < abs_gate, rec
    return rec


def _make_layer(N, K, nbits, gs, tdt, seed=0, zeros_kind="tensor", fma=True, scales_kind="group", out_dt=None):
    np_f = np.float16
    W_q, scales, zeros = O.gen_data(N, K, nbits, gs if scales_kind == "group" else K, seed=seed, np_float=np_f)
    W_q_t = torch.from_numpy(W_q).to(DEV)
    s_t = torch.from_numpy(scales.astype(np.float32)).to(tdt).to(DEV)
    z_t = torch.from_numpy(zeros.astype(np.float32)).to(tdt).to(DEV)
    code = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt]
    lin = GemLiteLinear(nbits, gs, K, N, code, code if out_dt is None else out_dt)
    zarg = {"tensor": z_t, "none": None, "int": (2 ** nbits) // 2}[zeros_kind]
    lin.pack(W_q_t, s_t, zarg, None, fma_mode=fma)
    return lin


def _oracle_from_layer(lin, x, scales_x=None):
    """Exact (float64) evaluation on the tensors the layer actually holds."""
    meta = lin.get_meta_args()
    e, wgm, csm = meta[4], meta[10], meta[9]
    s = O.to_f64(lin.scales.data) if lin.scales.numel() else None
    z = O.to_f64(lin.zeros.data).reshape(-1) if lin.zeros.numel() == 1 else (O.to_f64(lin.zeros.data) if lin.zeros.numel() else None)
    xf = O.to_f64(x).reshape(-1, x.shape[-1])
    if e > 1:
        pb = lin.W_q.element_size() * 8
        Wp = lin.W_q.data.cpu().numpy()
        N = Wp.shape[1]
        step = N if Wp.size <= (1 << 22) else 1024  # column blocks keep the float64 temporaries small
        outs = []
        for n0 in range(0, N, step):
            sl = slice(n0, n0 + step)
            s_b = s[..., sl] if (s is not None and s.ndim == 2) else s
            z_b = z[..., sl] if (z is not None and z.ndim == 2) else z
            outs.append(O.forward_packed(xf, Wp[:, sl], s_b, z_b, W_nbits=lin.W_nbits, group_size=lin.group_size,
                                         W_group_mode=wgm, channel_scale_mode=csm, scales_x=scales_x,
                                         zero_is_scalar=lin.zeros.numel() == 1, pack_bits=pb))
        return np.concatenate(outs, axis=1)
    W_kn = O.to_f64(lin.W_q.data)
    W = O.dequantize(W_kn, s if wgm >= 2 else None, z if wgm in (1, 3, 4) else None, lin.group_size, wgm,
                     lin.zeros.numel() == 1)
    return O.forward(xf, W, scales_w_channel=s if csm in (1, 3) else None, scales_x=scales_x, channel_scale_mode=csm)


# ------------------------------------------------------------------------------------------------ env
def test_device_is_mi355x_and_library_loaded():
    props = torch.cuda.get_device_properties(0)
    info = dict(name=props.name, cus=props.multi_processor_count, arch=getattr(props, "gcnArchName", "?"),
                lib=_hip.LIB_PATH, build=_hip.load().gemlite_hip_build_info().decode(), cpus=os.cpu_count())
    REPORT.append(dict(tag="env", **info))
    assert "gfx950" in info["arch"], info


# ------------------------------------------------------------------- golden vectors of the reference
CASES = load_cases()
FWD = [(c, M) for c in CASES for M in sorted(c["x"])]


@pytest.mark.parametrize("case,M", FWD, ids=[f"{c['name']}-M{M}" for c, M in FWD])
def test_golden_reference_outputs(case, M):
    """HIP output vs the outputs of the reference's own Triton kernels on the same packed tensors."""
    cfg, meta = case["cfg"], case["meta_args"]
    tdt = cfg["tdt"]
    lin = GemLiteLinear()
    sd = {"W_q": as_torch(case["W_q"], _code_of(case, "W_q")).to(DEV), "bias": None,
          "scales": as_torch(case["scales"], meta[8] if case["scales"].size else 6).to(DEV),
          "zeros": as_torch(case["zeros"], meta[8] if case["zeros"].size > 1 else 6).to(DEV),
          "metadata": torch.tensor(meta, dtype=torch.int32), "orig_shape": torch.tensor([cfg["N"], cfg["K"]], dtype=torch.int32)}
    if meta[4] == 1:  # unpacked weights are a transposed view of W[N, K]
        w_code = 1 if cfg["nb"] == 16 else (4 if cfg["in_dt"] == 4 else 3)
        sd["W_q"] = as_torch(case["W_in"], w_code).to(DEV).t()
    lin.load_state_dict(sd)
    x_code = meta[5] if not cfg["scaled_act"] else tdt
    x = as_torch(case["x"][M], x_code).to(DEV)
    refs = {mt: y for (mt, m), y in case["y"].items() if m == M}
    tried = 0
    for mt_name in ["AUTO", "GEMV_REVSPLITK", "GEMM_SPLITK", "GEMM"]:
        if mt_name.startswith("GEMV") and M > 1:
            continue
        y = lin(x) if mt_name == "AUTO" else lin.forward_manual(x, mt_name)
        torch.cuda.synchronize()
        kern = _kernel_name(lin, x, -1 if mt_name == "AUTO" else gemlite_amd.core.GEMLITE_MATMUL_TYPES_MAPPING[mt_name])
        # (1) against the exact oracle
        sx = None
        if cfg["scaled_act"]:
            _, sx = O.scale_activations_per_token(x, meta[5])
            xq, _ = O.scale_activations_per_token(x, meta[5])
            y_or = _oracle_from_layer(lin, torch.from_numpy(xq), sx)
        else:
            y_or = _oracle_from_layer(lin, x)
        fp8 = meta[5] in (3, 8)
        _compare(f"golden/{case['name']}/M{M}/{mt_name}/oracle", y, y_or, meta[6], abs_gate=5e-3 if fp8 else 1e-3,
                 extra=dict(kernel=kern))
        # (2) against what the reference's kernels produced (their fp16 accumulation noise included)
        for ref_mt, y_ref in refs.items():
            loose = 6.0 if ref_mt == "GEMV_REVSPLITK" else 2.5
            _compare(f"golden/{case['name']}/M{M}/{mt_name}/ref:{ref_mt}", y, y_ref, meta[6], abs_gate=None,
                     rel_tol=REL_TOL[meta[6]] * loose, extra=dict(kernel=kern))
        tried += 1
    assert tried >= 2


def _code_of(case, key):
    dt = case[key].dtype
    return {np.dtype("int32"): 6, np.dtype("uint8"): 5, np.dtype("int16"): 9, np.dtype("float16"): 1,
            np.dtype("int8"): 4, np.dtype("float32"): 0}[dt]


# ------------------------------------------------------------- north-star configurations, full size
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("M", [1, 2, 3, 4, 8, 16, 33, 64, 256])
def test_cfgA_a16w4_g128_4096(M, tdt):
    lin = _make_layer(4096, 4096, 4, 128, tdt, seed=0)
    assert lin.get_meta_args()[:5] == [0, 4, 128, 15, 8] and lin.W_group_mode == 4 and lin.data_contiguous
    x = torch.from_numpy(O.gen_x(M, 4096, seed=M).astype(np.float32)).to(tdt).to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    _compare(f"cfgA/{str(tdt)[6:]}/M{M}", y, _oracle_from_layer(lin, x), lin.output_dtype.value,
             extra=dict(kernel=_kernel_name(lin, x)))
    if M <= 8:
        for mt in ("GEMV", "GEMV_REVSPLITK", "GEMV_SPLITK", "GEMM_SPLITK", "GEMM"):
            y2 = lin.forward_manual(x, mt)
            _compare(f"cfgA/{str(tdt)[6:]}/M{M}/manual:{mt}", y2, _oracle_from_layer(lin, x), lin.output_dtype.value,
                     extra=dict(kernel=_kernel_name(lin, x, gemlite_amd.core.GEMLITE_MATMUL_TYPES_MAPPING[mt])))


def test_cfgB_a16w4_g128_8192_m256_bf16():
    lin = _make_layer(8192, 8192, 4, 128, torch.bfloat16, seed=3)
    x = torch.from_numpy(O.gen_x(256, 8192, seed=7).astype(np.float32)).to(torch.bfloat16).to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    _compare("cfgB/bf16/M256", y, _oracle_from_layer(lin, x), 2, extra=dict(kernel=_kernel_name(lin, x)))


@pytest.mark.parametrize("nbits,gs,N,K", [(2, 128, 2048, 4096), (1, 128, 1024, 4096), (8, 128, 1024, 2048),
                                           (4, 64, 1024, 2048), (4, 32, 512, 1024), (2, 64, 512, 1024),
                                           (4, 16, 256, 512), (4, 4096, 1024, 4096)])
@pytest.mark.parametrize("M", [1, 5, 24, 100])
def test_bit_widths_and_group_sizes(nbits, gs, N, K, M):
    tdt = torch.float16
    lin = _make_layer(N, K, nbits, gs, tdt, seed=nbits + gs)
    x = torch.from_numpy(O.gen_x(M, K, seed=M + 11).astype(np.float32)).to(tdt).to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    _compare(f"bits/w{nbits}g{gs}/{N}x{K}/M{M}", y, _oracle_from_layer(lin, x), 1, extra=dict(kernel=_kernel_name(lin, x)))


@pytest.mark.parametrize("zeros_kind,fma,scales_kind", [("tensor", False, "group"), ("none", True, "group"),
                                                         ("int", True, "group"), ("int", True, "channel"),
                                                         ("tensor", True, "channel"), ("none", True, "channel")])
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("M", [1, 4, 32, 128])
def test_all_weight_modes(zeros_kind, fma, scales_kind, tdt, M):
    lin = _make_layer(1024, 2048, 4, 128 if scales_kind == "group" else 2048, tdt, seed=5, zeros_kind=zeros_kind, fma=fma,
                      scales_kind=scales_kind)
    x = torch.from_numpy(O.gen_x(M, 2048, seed=M).astype(np.float32)).to(tdt).to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    _compare(f"modes/{zeros_kind}-{fma}-{scales_kind}/{str(tdt)[6:]}/M{M}/wgm{lin.W_group_mode}csm{lin.channel_scale_mode}", y,
             _oracle_from_layer(lin, x), lin.output_dtype.value, extra=dict(kernel=_kernel_name(lin, x)))


def test_config5_a16w2_16384_m1():
    lin = _make_layer(16384, 16384, 2, 128, torch.float16, seed=9)
    x = torch.from_numpy(O.gen_x(1, 16384, seed=2).astype(np.float32)).half().to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    _compare("cfg5/a16w2/16384/M1", y, _oracle_from_layer(lin, x), 1, extra=dict(kernel=_kernel_name(lin, x)))


# ----------------------------------------------------------------------- A8W8 / FP8 (config 4 path)
@pytest.mark.parametrize("M", [1, 16, 100, 256])
def test_a8w8_int8_dynamic_is_exact(M):
    torch.manual_seed(M)
    W = (torch.randn(4096, 4096) / 30).half()
    proc = gemlite_amd.helper.A8W8_int8_dynamic(device=DEV, dtype=torch.float16)
    lin = proc.from_weights(W)
    assert (lin.W_group_mode, lin.channel_scale_mode, lin.elements_per_sample) == (0, 3, 1)
    x = (torch.randn(M, 4096) / 10).half().to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    xq, sx = O.scale_activations_per_token(x, O.INT8)
    acc = xq @ O.to_f64(lin.W_q.data)  # exact integers
    y_or = acc * (sx.astype(np.float64) * O.to_f64(lin.scales.data).reshape(1, -1))
    _compare(f"a8w8/int8/M{M}", y, y_or, 1, extra=dict(kernel=_kernel_name(lin, x)))
    # activation quantisation itself: bit-exact vs the oracle
    xq_g, sx_g = scale_activations_per_token(x, torch.int8)
    assert np.array_equal(xq_g.cpu().numpy().astype(np.float64), xq) and np.array_equal(sx_g.cpu().numpy(), sx)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16, torch.float32], ids=["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("qdt", [torch.int8, torch.float8_e4m3fn, torch.float8_e5m2], ids=["int8", "e4m3", "e5m2"])
def test_activation_quantiser_register_and_fallback_paths_are_bit_exact(qdt, tdt):
    """Per-token quantiser (quant_utils.py:231-305): the row-in-registers kernel (16-bit x, K % 8 == 0, K <= 16384: 1 / 2 / 4 / 8
    16-byte loads per thread, ragged last load) and the scalar kernel everything else takes (fp32 x, K % 8 != 0, K > 16384, rows
    that are not 16-byte aligned) — codes and scales bit for bit against the oracle, rows of a strided view, an all-zero row, an
    outlier row."""
    code = {torch.int8: O.INT8, torch.float8_e4m3fn: O.FP8E4, torch.float8_e5m2: O.FP8E5}[qdt]
    g = torch.Generator().manual_seed(17)
    for K in (8, 64, 100, 2040, 2048, 2056, 4096, 11008, 16384, 16392):
        for M in (1, 5, 67):
            wide = (torch.randn(M, K + 24, generator=g) * torch.rand(M, 1, generator=g) * 4).to(tdt)
            if M > 1:
                wide[1] = 0
                wide[2, K // 2] = 900.0
            for x in (wide[:, :K].contiguous(), wide[:, :K], wide[:, 4:K + 4]):   # contiguous | row stride K + 24 | + rows off 16-byte alignment
                xq, sx = scale_activations_per_token(x.to(DEV) if x.is_contiguous() else wide.to(DEV)[:, x.storage_offset() % (K + 24):][:, :K], qdt)
                torch.cuda.synchronize()
                xq_o, sx_o = O.scale_activations_per_token(x.contiguous(), code)
                assert np.array_equal(sx.cpu().numpy().reshape(-1), sx_o.reshape(-1)), (K, M, "scales")
                assert np.array_equal(O.to_f64(xq), xq_o), (K, M, tuple(x.stride()), "codes")


@pytest.mark.parametrize("M", [1, 16, 200])
def test_fp8_fp8_dynamic(M):
    torch.manual_seed(M + 5)
    W = (torch.randn(2048, 4096) / 30).half()
    lin = gemlite_amd.helper.A8W8_fp8_dynamic(device=DEV, dtype=torch.float16).from_weights(W)
    x = (torch.randn(M, 4096) / 10).half().to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    xq, sx = O.scale_activations_per_token(x, O.FP8E4)
    y_or = (xq @ O.to_f64(lin.W_q.data)) * (sx.astype(np.float64) * O.to_f64(lin.scales.data).reshape(1, -1))
    _compare(f"a8w8/fp8/M{M}", y, y_or, 1, abs_gate=5e-3, extra=dict(kernel=_kernel_name(lin, x)))
    xq_g, sx_g = scale_activations_per_token(x, torch.float8_e4m3fn)
    assert np.array_equal(xq_g.float().cpu().numpy().astype(np.float64), xq) and np.array_equal(sx_g.cpu().numpy(), sx)


def test_fp16_fp16_unpacked():
    torch.manual_seed(0)
    W = (torch.randn(1024, 4096) / 30).half().to(DEV)
    lin = GemLiteLinear(16, None, 4096, 1024, DType.FP16, DType.FP16)
    lin.pack(W, None, None, None)
    assert lin.W_group_mode == 0 and lin.channel_scale_mode == 0 and lin.data_contiguous is False
    for M in (1, 4):
        x = (torch.randn(M, 4096) / 10).half().to(DEV)
        _compare(f"fp16xfp16/M{M}", lin(x), O.to_f64(x) @ O.to_f64(W).T, 1, extra=dict(kernel=_kernel_name(lin, x)))


# ----------------------------------------------------------------- integer work: bit-exact, on GPU
@pytest.mark.parametrize("nbits", [1, 2, 4, 8])
@pytest.mark.parametrize("pb", [8, 32])
def test_gpu_pack_unpack_bit_exact(nbits, pb):
    g = torch.Generator().manual_seed(nbits * 100 + pb)
    W = torch.randint(0, 2 ** nbits, (1024, 4096), generator=g, dtype=torch.int32).to(torch.uint8)
    cpu, e = bitpack.pack_weights_over_cols(W, nbits, pb, True)
    gpu, e2 = bitpack.pack_weights_over_cols(W.to(DEV), nbits, pb, True)
    assert e == e2 and gpu.dtype == cpu.dtype and torch.equal(gpu.cpu(), cpu.contiguous())
    assert np.array_equal(gpu.cpu().numpy(), O.pack_over_cols(W.numpy(), nbits, pb))
    back = bitpack.unpack_over_cols(gpu.t().contiguous(), nbits, 4096)
    assert torch.equal(back.cpu(), W)
    # round 5 (the tiled 32-bit pack kernel): ragged N and K (not multiples of its 64 x 256 tile) and a strided, unaligned view as the source
    Wr = torch.randint(0, 2 ** nbits, (1000, 2080 + 40), generator=g, dtype=torch.int32).to(torch.uint8)
    for view in (Wr[:, :2080].contiguous(), Wr[:, 3:3 + 2080]):
        cpu_r, _ = bitpack.pack_weights_over_cols(view.contiguous(), nbits, pb, True)
        gpu_r, _ = bitpack.pack_weights_over_cols(Wr.to(DEV)[:, 3:3 + 2080] if not view.is_contiguous() else view.to(DEV), nbits, pb, True)
        assert torch.equal(gpu_r.cpu(), cpu_r.contiguous()), (nbits, pb, view.is_contiguous())


# ------------------------------------------------------ properties at full size (size-independent)
def test_determinism_and_workspace_reset():
    lin = _make_layer(4096, 4096, 4, 128, torch.float16, seed=1)
    for M in (1, 4, 16):
        x = torch.from_numpy(O.gen_x(M, 4096, seed=3)).to(DEV)
        ys = [lin(x).clone() for _ in range(5)]
        torch.cuda.synchronize()
        assert all(torch.equal(ys[0], y) for y in ys[1:]), "split-K combine must be run-to-run deterministic"
    # split-K shapes on purpose (64-column tiles x 8 K slices at M = 1, the MFMA kernels' own split-K above)
    from gemlite_amd.core import _hip_matmul
    x1 = torch.from_numpy(O.gen_x(1, 4096, seed=3)).to(DEV)
    y_sk = [_hip_matmul(x1, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 1, (4, 8, 0, 0)).clone() for _ in range(3)]
    x256 = torch.from_numpy(O.gen_x(256, 4096, seed=4)).to(DEV)
    y_256 = [lin(x256).clone() for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(y_sk[0], y_sk[1]) and torch.equal(y_sk[0], y_sk[2])
    assert torch.equal(y_256[0], y_256[1]) and torch.equal(y_256[0], y_256[2])
    assert _hip._workspaces, "split-K launches must have allocated a workspace"
    ticket_words = 65536 - 4096  # the last 4096 words of the counter block belong to the opt-in timeline probes
    for (dev, stream), ws in _hip._workspaces.items():
        # arrival counters (and only they) must be back to zero; slabs may hold stale partial sums
        counters = ws[: ticket_words * 4].view(torch.int32)
        assert int(counters.abs().max().item()) == 0, f"split-K arrival counters not reset on stream {stream}"
    # a fresh launch after the loop still produces the same answer
    assert torch.equal(lin(x), ys[0])


def test_linearity_and_zero_input_full_size():
    lin = _make_layer(4096, 4096, 4, 128, torch.float16, seed=2)
    x1 = torch.from_numpy(O.gen_x(1, 4096, seed=1)).to(DEV)
    x2 = torch.from_numpy(O.gen_x(1, 4096, seed=2)).to(DEV)
    y1, y2, y12 = lin(x1).float(), lin(x2).float(), lin(x1 + x2).float()
    ref = O.to_f64(y1 + y2)
    _compare("prop/linearity", y12, ref, 1, rel_tol=3e-3)
    y0 = lin(torch.zeros_like(x1))
    assert float(y0.abs().max()) == 0.0
    # e_k probes: column k of the dequantised matrix, exact up to fp16 output rounding
    k = 1234
    ek = torch.zeros(1, 4096, dtype=torch.float16, device=DEV)
    ek[0, k] = 1.0
    _compare("prop/unit-vector", lin(ek), _oracle_from_layer(lin, ek), 1, rel_tol=1e-3)


def test_hip_graph_capture_of_decode_step():
    lin = _make_layer(4096, 4096, 4, 128, torch.float16, seed=4)
    x = torch.from_numpy(O.gen_x(1, 4096, seed=8)).to(DEV)
    y_eager = lin(x).clone()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            lin(x)  # warm-up on the capture stream: allocates its workspace
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        y_static = lin(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_static, y_eager)


def test_bias_and_batched_input_shapes():
    lin = _make_layer(1024, 2048, 4, 128, torch.float16, seed=6)
    bias = torch.randn(1024).half().to(DEV)
    lin.bias = torch.nn.Parameter(bias, requires_grad=False)
    x = (torch.randn(2, 3, 2048) / 10).half().to(DEV)
    y = lin(x)
    assert y.shape == (2, 3, 1024)
    y_or = _oracle_from_layer(lin, x.reshape(-1, 2048)) + O.to_f64(bias).reshape(1, -1)
    _compare("bias/batched", y.reshape(-1, 1024), y_or, 1, rel_tol=2e-3)


def test_unsupported_raises_not_falls_back():
    lin = _make_layer(1024, 2048, 4, 128, torch.float16, seed=6)
    with pytest.raises(_hip.GemliteHipError):
        lin(torch.randn(1, 2048).half())  # CPU tensor
    with pytest.raises(ValueError):
        lin(torch.randn(1, 1024).half().to(DEV))  # wrong K


# --------------------------------------------------------------- odd shapes / kernel-boundary cases
@pytest.mark.parametrize("N,K,gs", [(4160, 4096, 128), (1000, 2048, 128), (4096, 4224, 64), (2048, 14336, 128),
                                      (14336, 4096, 128), (6144, 4096, 128), (128, 512, 32), (64, 64, 64)])
@pytest.mark.parametrize("M", [1, 2, 7, 17, 65, 129])
def test_odd_shapes(N, K, gs, M):
    """Shapes off the tuned grid: every one must land on SOME native kernel and match the oracle."""
    tdt = torch.float16
    lin = _make_layer(N, K, 4, gs, tdt, seed=N % 97 + K % 89)
    x = torch.from_numpy(O.gen_x(M, K, seed=M + 3).astype(np.float32)).to(tdt).to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    _compare(f"odd/{N}x{K}g{gs}/M{M}", y, _oracle_from_layer(lin, x), 1, extra=dict(kernel=_kernel_name(lin, x)))


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [2, 1])
@pytest.mark.parametrize("M", [1, 16, 48, 200])
def test_low_bits_all_kernel_families(nbits, M, tdt):
    lin = _make_layer(2048, 4096, nbits, 128, tdt, seed=nbits)
    x = torch.from_numpy(O.gen_x(M, 4096, seed=M).astype(np.float32)).to(tdt).to(DEV)
    for mt in (None, "GEMM_SPLITK", "GEMM"):
        y = lin(x) if mt is None else lin.forward_manual(x, mt)
        torch.cuda.synchronize()
        _compare(f"lowbits/w{nbits}/{str(tdt)[6:]}/M{M}/{mt}", y, _oracle_from_layer(lin, x), lin.output_dtype.value,
                 extra=dict(kernel=_kernel_name(lin, x, -1 if mt is None else gemlite_amd.core.GEMLITE_MATMUL_TYPES_MAPPING[mt])))


@pytest.mark.parametrize("zeros_kind,fma,scales_kind", [("tensor", False, "group"), ("none", True, "group"),
                                                         ("int", True, "group"), ("int", True, "channel"),
                                                         ("tensor", True, "channel")])
def test_tiled_kernel_all_modes_m256(zeros_kind, fma, scales_kind):
    tdt = torch.bfloat16
    lin = _make_layer(2048, 4096, 4, 128 if scales_kind == "group" else 4096, tdt, seed=7, zeros_kind=zeros_kind, fma=fma,
                      scales_kind=scales_kind)
    x = torch.from_numpy(O.gen_x(256, 4096, seed=5).astype(np.float32)).to(tdt).to(DEV)
    bias = torch.randn(2048).to(tdt).to(DEV)
    lin.bias = torch.nn.Parameter(bias, requires_grad=False)
    y = lin(x)
    torch.cuda.synchronize()
    y_or = _oracle_from_layer(lin, x) + O.to_f64(bias).reshape(1, -1)
    assert _kernel_name(lin, x).startswith("gemm_w4_mma")
    # |y| ~ 0.8 because of the bias: the absolute gate of the bias-free fixtures does not apply, the relative one does
    _compare(f"tiled-modes/{zeros_kind}-{fma}-{scales_kind}", y, y_or, 2, abs_gate=None, extra=dict(kernel="gemm_w4_mma_kernel"))


def test_split_k_variants_agree_bitwise_independent_of_run_order():
    """Forced split-K factors: each is deterministic and all agree with the oracle."""
    lin = _make_layer(4096, 4096, 4, 128, torch.float16, seed=11)
    from gemlite_amd.core import _hip_matmul
    for M, mt, tunings in ((1, 1, [(4, 1, 0, 0), (4, 2, 0, 0), (4, 8, 0, 0), (3, 4, 0, 0), (2, 1, 0, 1),
                                   (2, 1, 4, 0), (2, 1, 8, 0), (2, 1, 16, 0), (2, 1, 82, 0)]),  # waves per block
                           (16, 3, [(0, 1, 1, 0), (0, 2, 1, 0), (0, 8, 1, 0), (1, 1, 0, 0), (2, 2, 0, 0), (4, 4, 0, 0)]),
                           (256, 4, [(0, 1, 0, 0), (0, 4, 0, 0), (0, 16, 0, 0), (0, 32, 0, 0), (0, 2, 4, 0)])):
        x = torch.from_numpy(O.gen_x(M, 4096, seed=M)).to(DEV)
        y_or = _oracle_from_layer(lin, x)
        for t in tunings:
            ys = [_hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), mt, t).clone() for _ in range(3)]
            torch.cuda.synchronize()
            assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), (M, t)
            _compare(f"splitk/M{M}/{t}", ys[0], y_or, 1)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits,N,K", [(4, 4096, 4096), (2, 2048, 4096), (4, 1024, 8192)])
def test_direct_mfma_kernel_tiles_and_splits(nbits, N, K, tdt):
    """Registers-only MFMA kernel for few rows: every tile width, forced split-K, both row-tile counts, and the
    LDS-staged streaming kernel (tuning[2] = 1) as a cross-check — all against the oracle."""
    from gemlite_amd.core import _hip_matmul
    lin = _make_layer(N, K, nbits, 128, tdt, seed=21 + nbits)
    for M in (1, 2, 13, 16, 17, 32):
        x = torch.from_numpy(O.gen_x(M, K, seed=M + 3).astype(np.float32)).to(tdt).to(DEV)
        y_or = _oracle_from_layer(lin, x)
        for tuning in ((1, 0, 0, 0), (2, 0, 0, 0), (4, 0, 0, 0), (1, 2, 0, 0), (4, 2, 0, 0), (0, 0, 0, 0), (0, 0, 1, 0)):
            if tuning[0] == 1 and M > 16:
                continue  # the 16-column tile exists for one row tile only
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 3, tuning)
            torch.cuda.synchronize()
            _compare(f"direct/w{nbits}/{N}x{K}/{str(tdt)[6:]}/M{M}/{tuning}", y, y_or, lin.output_dtype.value)
    x = torch.from_numpy(O.gen_x(8, K, seed=1).astype(np.float32)).to(tdt).to(DEV)
    # (round 5: at 4096^2 from 8 rows the default is the decode-shaped rows kernel; tuning[3] & 65536 = the round-4 choice)
    assert _kernel_name(lin, x, -1, (0, 0, 0, 65536)).startswith("gemm_wn_direct_kernel"), _kernel_name(lin, x)


@pytest.mark.parametrize("zeros_kind,fma,scales_kind", [("tensor", False, "group"), ("none", True, "group"),
                                                         ("int", True, "group"), ("int", True, "channel"),
                                                         ("tensor", True, "channel"), ("none", True, "channel")])
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_direct_mfma_kernel_all_modes(zeros_kind, fma, scales_kind, tdt):
    lin = _make_layer(2048, 4096, 4, 128 if scales_kind == "group" else 4096, tdt, seed=5, zeros_kind=zeros_kind, fma=fma,
                      scales_kind=scales_kind)
    for M in (3, 6, 24):   # (3 rows: the MFMA GEMV since round 3 — the same modes on that kernel)
        x = torch.from_numpy(O.gen_x(M, 4096, seed=M).astype(np.float32)).to(tdt).to(DEV)
        y = lin(x)
        torch.cuda.synchronize()
        # (round 5: 16 .. 32 rows of a layer with >= 128 16-column tiles run the rows kernel by default; tuning[3] & 65536 = the round-4 choice)
        assert _kernel_name(lin, x).startswith(("gemm_wn_direct_kernel", "gemm_w4_rows_kernel") if M > 4 else "gemv_mfma_kernel"), _kernel_name(lin, x)
        _compare(f"direct-modes/{zeros_kind}-{fma}-{scales_kind}/{str(tdt)[6:]}/M{M}", y, _oracle_from_layer(lin, x),
                 lin.output_dtype.value)
        if M > 4:
            assert _kernel_name(lin, x, -1, (0, 0, 0, 65536)).startswith("gemm_wn_direct_kernel"), _kernel_name(lin, x, -1, (0, 0, 0, 65536))
            gemlite_amd.core.TUNING_OVERRIDE = (0, 0, 0, 65536)
            try:
                y4 = lin(x)
            finally:
                gemlite_amd.core.TUNING_OVERRIDE = None
            torch.cuda.synchronize()
            _compare(f"direct-modes-r4/{zeros_kind}-{fma}-{scales_kind}/{str(tdt)[6:]}/M{M}", y4, _oracle_from_layer(lin, x), lin.output_dtype.value)


def test_a8w8_mfma_kernel_matches_streaming_kernel_and_is_selected():
    """int8 x int8: the MFMA kernel and the sdot4 streaming kernel accumulate exactly, so they agree bit for bit."""
    torch.manual_seed(3)
    W = (torch.randn(2048, 4096) / 30).half()
    lin = gemlite_amd.helper.A8W8_int8_dynamic(device=DEV, dtype=torch.float16).from_weights(W)
    x = (torch.randn(77, 4096) / 10).half().to(DEV)
    # (round 4: 65 .. 256 rows whose 64 x 64 tiles fit one round of CUs run the unsplit kernel; the 128-row tile stays behind tuning[0] = 6)
    assert _kernel_name(lin, torch.empty(77, 4096, dtype=torch.int8)) == "gemm_a8w8_sq_kernel<64x64>", _kernel_name(lin, x)
    assert _kernel_name(lin, torch.empty(77, 4096, dtype=torch.int8), -1, (6, 0, 0, 0)) == "gemm_a8w8_lds_kernel<128x128>"
    y = lin(x)
    outs = {}
    # streaming kernel | 4-wave MFMA kernel of round 1 | 8-wave kernels: every tile height, K split 1 / 3 (uneven) / 8, weights
    # through LDS (128 / 256 rows, default) and straight from memory (tuning[3] & 64)
    for t in ((1, 0, 0, 0), (2, 0, 0, 0), (0, 1, 1, 0), (0, 3, 2, 0), (0, 8, 4, 0), (0, 1, 8, 0), (0, 5, 8, 0), (0, 3, 4, 0),
              (0, 8, 4, 64), (0, 1, 8, 64), (0, 5, 8, 64), (6, 0, 0, 0), (5, 0, 2, 0), (5, 0, 3, 0), (5, 0, 4, 0), (10, 0, 0, 0)):
        gemlite_amd.core.TUNING_OVERRIDE = t
        try:
            outs[t] = lin(x)
        finally:
            gemlite_amd.core.TUNING_OVERRIDE = None
    torch.cuda.synchronize()
    for t, y2 in outs.items():
        assert torch.equal(y, y2), t
    for M in (2, 5, 16, 17, 31, 40, 64):  # few rows: 16-column blocks with 1 / 2 / 4 row tiles (round 3: up to 64 rows at this size)
        xs = (torch.randn(M, 4096) / 10).half().to(DEV)
        want = "a8w8_rows_lds_kernel<%dx16>" % (16 if M <= 16 else (32 if M <= 32 else (48 if M <= 48 else 64)))   # (round 6; rounds 3-5: a8w8_rows_kernel)
        assert _kernel_name(lin, torch.empty(M, 4096, dtype=torch.int8)) == want
        ya = lin(xs)
        gemlite_amd.core.TUNING_OVERRIDE = (1, 0, 0, 0)
        try:
            yb = lin(xs)
        finally:
            gemlite_amd.core.TUNING_OVERRIDE = None
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), M


@pytest.mark.parametrize("kind", ["int8", "fp8e4", "fp8e5"])
def test_a8w8_sq128_kernel_unsplit_128_tiles(kind):
    """gemm_a8w8_sq_kernel<128x128> (round 5, tuning[0] = 10; the default where its tiles number about one per CU): ragged row tiles,
    column-tile counts that are / are not a multiple of the 8 XCDs, one .. many K steps, strided activations.  int8 accumulates exactly:
    bit for bit against the streaming kernel; fp8 sums in another order: 2 % of mean |y|."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    qdt = dict(int8=torch.int8, fp8e4=torch.float8_e4m3fn, fp8e5=torch.float8_e5m2)[kind]
    for (N, K) in ((1024, 256), (1152, 768), (2048, 4096), (128, 2048)):
        torch.manual_seed(N + K)
        W = (torch.randn(N, K) / 30).half()
        proc = H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16) if kind == "int8" else H.A8W8_dynamic(device=DEV, dtype=torch.float16, fp8=qdt)
        lin = proc.from_weights(W)
        for M in (2, 65, 128, 129, 300):
            x = (torch.randn(M, K) / 10).half().to(DEV)
            xq, sx = scale_activations_per_token(x, qdt)
            wide = torch.zeros(M, K + 64, dtype=xq.dtype, device=DEV)
            wide[:, :K] = xq
            outs = {}
            for label, (xin, t) in dict(sq128=(xq, (10, 0, 0, 0)), sq128_strided=(wide[:, :K], (10, 0, 0, 0)), stream=(xq, (1, 0, 0, 0)),
                                        sq128_5x128=(xq, (10, 0, 5, 0))).items():
                outs[label] = _hip_matmul(xin, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, t)
            torch.cuda.synchronize()
            assert _kernel_name(lin, xq, -1, (10, 0, 0, 0)) == "gemm_a8w8_sq_kernel<128x128>"
            assert torch.equal(outs["sq128"], outs["sq128_strided"]), (kind, N, K, M)
            assert torch.equal(outs["sq128"], outs["sq128_5x128"]), (kind, N, K, M)   # (same 128-byte steps, another stage count: bitwise)
            if kind == "int8":
                assert torch.equal(outs["sq128"], outs["stream"]), (kind, N, K, M)
            else:
                d = (outs["sq128"].float() - outs["stream"].float()).abs()
                assert float(d.max()) < 0.02 * float(outs["stream"].float().abs().mean()) + 1e-3, (kind, N, K, M, float(d.max()))


@pytest.mark.parametrize("K", [256, 512, 768, 1280])
def test_a8w8_lds_kernel_short_and_uneven_k(K):
    """Weights-through-LDS kernel with fewer K steps than LDS stages (K = 256: two 128-byte steps, 3-4 stages) and step counts
    that are not a multiple of the unrolled group; int8 is exact, so the streaming kernel is the reference, bit for bit."""
    torch.manual_seed(K)
    W = (torch.randn(256, K) / 30).half()
    lin = gemlite_amd.helper.A8W8_int8_dynamic(device=DEV, dtype=torch.float16).from_weights(W)
    for M in (70, 130, 300):
        x = (torch.randn(M, K) / 10).half().to(DEV)
        assert _kernel_name(lin, torch.empty(M, K, dtype=torch.int8), -1, (6, 0, 0, 0)) == "gemm_a8w8_lds_kernel<128x128>"
        if K % 256 == 0:  # the unsplit 64 x 64 tiles of round 4 (256-byte K steps): fewer steps than stages at K = 256 / 512 / 768
            assert _kernel_name(lin, torch.empty(M, K, dtype=torch.int8)) == "gemm_a8w8_sq_kernel<64x64>"
        outs = {}
        for t in (None, (6, 0, 0, 0), (0, 0, 8, 0), (0, 2, 4, 0), (1, 0, 0, 0)) + (((5, 0, 2, 0), (5, 0, 4, 0), (10, 0, 0, 0)) if K % 256 == 0 else ()):
            gemlite_amd.core.TUNING_OVERRIDE = t
            try:
                outs[t] = lin(x)
            finally:
                gemlite_amd.core.TUNING_OVERRIDE = None
        torch.cuda.synchronize()
        for t, y in outs.items():
            assert torch.equal(y, outs[(1, 0, 0, 0)]), (K, M, t)


@pytest.mark.parametrize("kind", ["int8", "fp8e4", "fp8e5"])
def test_a8w8_rows_kernel(kind):
    """2..64 rows of A8W8 (BASELINE config 4, M = 16): 16-column blocks, one 16-row MFMA per 64-k chunk and row tile — every M, K = 64 * odd
    (uneven chunk counts per wave), long K (several ring passes), against the oracle on the kernel's own quantised inputs, and
    bit-exact against the 8-wave MFMA kernel for int8."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    code = {"int8": O.INT8, "fp8e4": O.FP8E4, "fp8e5": O.FP8E5}[kind]
    qdt = {"int8": torch.int8, "fp8e4": torch.float8_e4m3fn, "fp8e5": torch.float8_e5m2}[kind]
    for (N, K) in ((512, 4096), (48, 64 * 37), (256, 16384)):
        g = torch.Generator().manual_seed(N + K)
        W = (torch.randn(N, K, generator=g) / 30).half()
        proc = H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16) if kind == "int8" else \
            H.A8W8_dynamic(device=DEV, dtype=torch.float16, fp8=qdt)
        lin = proc.from_weights(W)
        for M in (2, 3, 7, 16, 17, 29, 32, 33, 50, 64) + ((1,) if N == 512 else ()):
            x = (torch.randn(M, K, generator=g) / 10).half().to(DEV)
            xq, sx = scale_activations_per_token(x, qdt)
            tun = (4 if (M == 1 or M > 16) else 0, 0, 0, 524288)   # (forced: the x re-read rule sends 256 x 16384 at M > 21 elsewhere; & 524288: round 6 moved these shapes to gemm_w8_rows.hip)
            name = _kernel_name(lin, xq, -1, tun)
            assert name == ("a8w8_rows_kernel<16x16>" if M <= 16 else ("a8w8_rows_kernel<32x16>" if M <= 32 else "a8w8_rows_kernel<64x16>")), (M, name)
            y = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, tun)
            torch.cuda.synchronize()
            xq_o, sx_o = O.scale_activations_per_token(x, code)
            assert np.array_equal(O.to_f64(xq), xq_o), "activation quantiser differs from the oracle"
            y_or = (xq_o @ O.to_f64(lin.W_q.data)) * (sx_o.astype(np.float64) * O.to_f64(lin.scales.data).reshape(1, -1))
            _compare(f"a8w8_rows/{kind}/{N}x{K}/M{M}", y, y_or, 1, abs_gate=5e-3, extra=dict(kernel=name))
            if kind == "int8" and N % 128 == 0:
                y2 = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, (0, 1, 1, 0))
                torch.cuda.synchronize()
                assert torch.equal(y, y2), (N, K, M)


@pytest.mark.parametrize("M", [1, 64])
def test_fp8_e5m2_dynamic(M):
    """e5m2 x e5m2 (FP8e5, dtype code 8): streaming kernel at M = 1, v_mfma_f32_16x16x32_bf8_bf8 (16-column blocks, 4 row tiles) at M = 64."""
    torch.manual_seed(M + 9)
    W = (torch.randn(1024, 2048) / 30).half()
    lin = gemlite_amd.helper.A8W8_dynamic(device=DEV, dtype=torch.float16, fp8=torch.float8_e5m2).from_weights(W)
    x = (torch.randn(M, 2048) / 10).half().to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    xq, sx = O.scale_activations_per_token(x, O.FP8E5)
    y_or = (xq @ O.to_f64(lin.W_q.data)) * (sx.astype(np.float64) * O.to_f64(lin.scales.data).reshape(1, -1))
    _compare(f"a8w8/fp8e5/M{M}", y, y_or, 1, abs_gate=5e-3)


def test_autotune_layer_fills_the_tuning_table_and_results_stay_correct():
    from gemlite_amd import core
    core.GemLiteLinear.reset_config()
    lin = _make_layer(2048, 4096, 4, 128, torch.float16, seed=31)
    res = gemlite_amd.helper.autotune_layer(lin, batch_sizes=(1, 8, 256), iters=5)
    assert set(res) == {1, 8, 256} and all(len(v["tuning"]) == 4 and v["us"] > 0 for v in res.values())
    try:
        for M in (1, 8, 256):
            x = torch.from_numpy(O.gen_x(M, 4096, seed=M)).to(DEV)
            a = core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
            assert core.lookup_tuning(-1, M, a) == tuple(res[M]["tuning"])
            y = lin(x)
            torch.cuda.synchronize()
            _compare(f"autotuned/M{M}/{res[M]['tuning']}", y, _oracle_from_layer(lin, x), 1)
    finally:
        core.GemLiteLinear.reset_config()


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [4, 2])
def test_direct_mfma_kernel_group_size_64(nbits, tdt):
    """Group size 64 (hqq's default): scale / zero folded in every 64 k; registers-only kernel for one row tile."""
    from gemlite_amd.core import _hip_matmul
    lin = _make_layer(2048, 4096, nbits, 64, tdt, seed=41 + nbits)
    for M in (2, 9, 16):
        x = torch.from_numpy(O.gen_x(M, 4096, seed=M + 5).astype(np.float32)).to(tdt).to(DEV)
        y_or = _oracle_from_layer(lin, x)
        assert _kernel_name(lin, x, -1, (0, 0, 0, 65536)).startswith("gemm_wn_direct_kernel" if M > 4 else ("gemv_mfma_kernel", "gemm_wn_direct_kernel")), _kernel_name(lin, x)
        for tuning in ((0, 0, 0, 0), (0, 0, 0, 65536), (1, 0, 0, 0), (2, 0, 0, 0), (4, 0, 0, 0), (4, 2, 0, 0), (0, 0, 1, 0)):
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tuning)
            torch.cuda.synchronize()
            _compare(f"direct-g64/w{nbits}/{str(tdt)[6:]}/M{M}/{tuning}", y, y_or, lin.output_dtype.value)
    # 17 .. 32 rows: round 5 sends them to the rows kernel (where it does not pay: the 32-row MFMA tiles), no longer to the LDS-staged streaming kernel
    x = torch.from_numpy(O.gen_x(24, 4096, seed=3).astype(np.float32)).to(tdt).to(DEV)
    name = _kernel_name(lin, x)
    assert name.startswith("gemm_w4_rows_kernel" if nbits == 4 else "gemm_w2_rows_kernel"), name
    assert _kernel_name(lin, x, -1, (3, 0, 0, 65536)).startswith("gemm_w%d_mma_kernel<32x128>" % nbits)
    for tuning in ((0, 0, 0, 0), (3, 0, 0, 65536), (0, 0, 0, 65536)):
        y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tuning)
        torch.cuda.synchronize()
        _compare(f"direct-g64/w{nbits}/{str(tdt)[6:]}/M24/{tuning}", y, _oracle_from_layer(lin, x), lin.output_dtype.value)


@pytest.mark.parametrize("M", [1, 8])
def test_remaining_helper_processors_against_the_oracle(M):
    """A16W8 (int8 / fp8 weight-only, pre- and post-scale), A8W4 dynamic (fp8 activations x 4-bit groups) and the two
    BitNet processors: whatever kernel they land on, the result matches the float64 evaluation of the stored tensors."""
    H = gemlite_amd.helper
    torch.manual_seed(11 + M)
    N, K = 1024, 2048
    W = (torch.randn(N, K) / 30).half()
    x = (torch.randn(M, K) / 10).half().to(DEV)
    for name, lin in (("a16w8-pre", H.A16W8(device=DEV).from_weights(W)),
                      ("a16w8-post", H.A16W8(device=DEV, post_scale=True).from_weights(W)),
                      ("a16w8-fp8", H.A16W8_FP8(device=DEV).from_weights(W))):
        y = lin(x)
        torch.cuda.synchronize()
        assert _kernel_name(lin, x).startswith(("a16w8_rows_kernel", "a16w8_rows_lds_kernel", "a16w8_decode_kernel")), _kernel_name(lin, x)
        _compare(f"helpers/{name}/M{M}", y, _oracle_from_layer(lin, x), 1, abs_gate=5e-3, extra=dict(kernel=_kernel_name(lin, x)))
    W_q, sc, zr = O.gen_data(N, K, 4, 128, seed=3)
    Wt = torch.randint(-1, 2, (N, K)).half()
    for name, lin, code in (
            ("a8w4-dyn", H.A8W4_HQQ_INT_dynamic(device=DEV).from_weights(torch.from_numpy(W_q), torch.from_numpy(sc), torch.from_numpy(zr)), O.FP8E4),
            ("a8w158-dyn", H.A8W158_INT_dynamic(device=DEV).from_weights(Wt, torch.tensor(0.02)), O.INT8)):
        y = lin(x)
        torch.cuda.synchronize()
        xq, sx = O.scale_activations_per_token(x, code)
        # fp8 activations: the dequantised weights are rounded to fp8 before the dot, like the reference (gemm_kernels.py:384)
        y_or = O.forward_packed(xq, lin.W_q.data.cpu().numpy(), O.to_f64(lin.scales.data), O.to_f64(lin.zeros.data).reshape(-1)
                                if lin.zeros.numel() == 1 else O.to_f64(lin.zeros.data), W_nbits=lin.W_nbits, group_size=lin.group_size,
                                W_group_mode=lin.W_group_mode, channel_scale_mode=lin.channel_scale_mode, scales_x=sx,
                                zero_is_scalar=lin.zeros.numel() == 1,
                                # M = 1 is the reference's GEMV family: the dequantised weight stays in the metadata type there (fp16
                                # here; gemv_revsplitK_kernels.py:331-332, pinned by tests/golden/fullsize_ref_r4.npz a8w4_fp8dyn_m1)
                                weight_cast_code=(O.FP8E4 if M > 1 else O.FP16) if code == O.FP8E4 else None)
        kname = _kernel_name(lin, x)
        want = {(O.FP8E4, 1): "gemv_a8w4_kernel<tile16,16w>", (O.FP8E4, 8): "a8w4_rows_kernel<16x16>",
                (O.INT8, 1): "gemv_a8w2_kernel<tile16,16w>", (O.INT8, 8): "a8w2_rows_kernel<16x16>"}[(code, M)]
        assert kname == want, kname
        _compare(f"helpers/{name}/M{M}", y, y_or, 1, abs_gate=5e-3, extra=dict(kernel=kname))
    lin = H.A16W158_INT(device=DEV).from_weights(Wt, torch.tensor(0.02))
    y = lin(x)
    torch.cuda.synchronize()
    # fp32 channel scale: at M = 1 the GEMV family (its epilogue reads any float scale type, round 4), above it the untyped epilogue of the tile kernel
    # (late round 5: from 8 rows of this narrow layer the rows kernel, whose epilogue reads fp32 channel scales too; tuning[3] & 65536 = round 4)
    assert _kernel_name(lin, x).startswith(("gemv_wn_kernel<tile", "gemv_w2_mfma_kernel<tile") if M == 1 else ("gemm_wn_direct_kernel<tile", "gemm_w2_rows_kernel<")), _kernel_name(lin, x)
    if M > 1:
        assert _kernel_name(lin, x, -1, (0, 0, 0, 65536)).startswith("gemm_wn_direct_kernel<tile"), _kernel_name(lin, x, -1, (0, 0, 0, 65536))
    # ... and the tile kernel's untyped epilogue above 32 rows
    x64 = (torch.randn(64, K) / 10).half().to(DEV)
    assert _kernel_name(lin, x64, -1, (0, 0, 0, 65536)).startswith("gemm_w2_mma_kernel<"), _kernel_name(lin, x64, -1, (0, 0, 0, 65536))
    _compare(f"helpers/a16w158/M64", lin(x64), _oracle_from_layer(lin, x64), 1)
    gemlite_amd.core.TUNING_OVERRIDE = (0, 0, 0, 65536)
    try:
        y64_r4, y_r4 = lin(x64), lin(x)
    finally:
        gemlite_amd.core.TUNING_OVERRIDE = None
    torch.cuda.synchronize()
    _compare(f"helpers/a16w158/M64/r4", y64_r4, _oracle_from_layer(lin, x64), 1)
    _compare(f"helpers/a16w158/M{M}/r4", y_r4, _oracle_from_layer(lin, x), 1)
    _compare(f"helpers/a16w158/M{M}", y, _oracle_from_layer(lin, x), 1, extra=dict(kernel=_kernel_name(lin, x)))


# ------------------------------------------------------------------------------------------ round 2 additions
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [4, 2, 1, 8])
def test_mma_kernel_bit_widths_tiles_and_splits(nbits, tdt):
    """8-wave MFMA kernel: every bit width x every tile height (tuning[2] = rows / 32), ragged M, several M tiles,
    K steps of 128 and 256, forced split-K — all against the oracle."""
    from gemlite_amd.core import _hip_matmul
    N, K = 256, 1280
    lin = _make_layer(N, K, nbits, 128, tdt, seed=40 + nbits)
    for mi, M in ((1, 29), (2, 64), (4, 100), (8, 256), (8, 300), (4, 130)):
        x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
        y_or = _oracle_from_layer(lin, x)
        for sk in (0, 1, 3, 4):  # 3 / 4 slices of 5 or 10 steps: uneven K slices
            tuning = (0, sk, mi, 0)
            name = _kernel_name(lin, x, 4, tuning)
            assert name.startswith(f"gemm_w{nbits}_mma_kernel<{32 * mi}x128>"), name
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, tuning)
            torch.cuda.synchronize()
            # 8-bit codes: |y| grows with the code range, the absolute gate of the 4-bit fixtures does not apply
            _compare(f"mma/w{nbits}/{str(tdt)[6:]}/M{M}/mi{mi}/sk{sk}", y, y_or, lin.output_dtype.value,
                     abs_gate=None if nbits == 8 else 1e-3, extra=dict(kernel=name))


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [4, 2])
def test_mma_kernel_wide_tiles(nbits, tdt):
    """The 256-column tiles of the 8-wave MFMA kernel (tuning[2] = 16 + rows / 32): 128 / 256 rows, ragged M, several M and
    N tiles, 64-k steps (K = 64 * 21: uneven slices), forced split-K, group sizes 128 and 64 — against the oracle."""
    from gemlite_amd.core import _hip_matmul
    for (N, K, gs) in ((512, 1344, 64), (256, 1280, 128)):
        lin = _make_layer(N, K, nbits, gs, tdt, seed=70 + nbits)
        for mi, M in ((4, 100), (4, 130), (8, 256), (8, 300)):
            x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
            y_or = _oracle_from_layer(lin, x)
            for sk in (0, 1, 2, 3):
                tuning = (0, sk, 16 + mi, 0)
                name = _kernel_name(lin, x, 4, tuning)
                assert name == f"gemm_w{nbits}_mma_kernel<{32 * mi}x256>", name
                y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, tuning)
                torch.cuda.synchronize()
                _compare(f"mma_wide/w{nbits}/{str(tdt)[6:]}/{N}x{K}/M{M}/mi{mi}/sk{sk}", y, y_or, lin.output_dtype.value,
                         abs_gate=1e-3, extra=dict(kernel=name))


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [4, 2])
def test_mma_kernel_narrow_tiles(nbits, tdt):
    """Round 4: the 64-column tiles of the 8-wave MFMA kernel (tuning[2] = 32 + variant; four K quarters per step inside the block,
    K normally NOT split over blocks): 64 / 128 rows, 256- and 512-k steps, 2 / 3 LDS stages, ragged M, several M and N tiles, uneven
    forced K slices, group sizes 128 and 64, few K steps (K = 512: 1 or 2 steps, fewer than the stages / the register ring) — all
    against the oracle."""
    from gemlite_amd.core import _hip_matmul
    for (N, K, gs) in ((192, 1536, 64), (256, 2560, 128), (128, 512, 128)):
        lin = _make_layer(N, K, nbits, gs, tdt, seed=75 + nbits)
        for M in (29, 64, 100, 130, 300):
            x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
            y_or = _oracle_from_layer(lin, x)
            for v in (0, 1, 2, 3):
                for sk in (0, 1, 2, 3):
                    tuning = (0, sk, 32 + v, 0)
                    try:
                        name = _kernel_name(lin, x, 4, tuning)
                    except Exception:
                        continue  # more slices than K steps
                    if not name.startswith(f"gemm_w{nbits}_mma_kernel<{64 if v < 2 else 128}x64>"):
                        assert sk > (K // (512 if v == 1 else 256)), (name, tuning)  # only an impossible split may fall elsewhere
                        continue
                    y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, tuning)
                    torch.cuda.synchronize()
                    _compare(f"mma_narrow/w{nbits}/{str(tdt)[6:]}/{N}x{K}/M{M}/v{v}/sk{sk}", y, y_or, lin.output_dtype.value,
                             abs_gate=1e-3, extra=dict(kernel=name))


def test_mma_kernel_narrow_tiles_all_modes_and_headline_choice():
    """Every W_group_mode through the narrow tiles, and the planner's choice at the headline shape (cfgA, M = 256: 64 x 64 tiles, K
    unsplit — 20.3 -> 17.4 us, profiles/r04/probe_mma_narrow_*.log)."""
    from gemlite_amd.core import _hip_matmul
    for mode, kw in (("fma", {}), ("sub_mul", dict(fma=False)), ("symmetric", dict(zeros_kind="none")), ("int_zero", dict(zeros_kind="int"))):
        lin = _make_layer(256, 1024, 4, 128, torch.bfloat16, seed=83, **kw)
        x = torch.from_numpy(O.gen_x(200, 1024, seed=5).astype(np.float32)).to(torch.bfloat16).to(DEV)
        y_or = _oracle_from_layer(lin, x)
        for v in (0, 2):
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, (0, 0, 32 + v, 0))
            torch.cuda.synchronize()
            _compare(f"mma_narrow/modes/{mode}/v{v}", y, y_or, lin.output_dtype.value, abs_gate=1e-3)
    lin = _make_layer(4096, 4096, 4, 128, torch.bfloat16, seed=84)
    x = torch.from_numpy(O.gen_x(256, 4096, seed=6).astype(np.float32)).to(torch.bfloat16).to(DEV)
    assert _kernel_name(lin, x) == "gemm_w4_mma_kernel<64x64>", _kernel_name(lin, x)


@pytest.mark.parametrize("bits", [4, 2])
def test_narrow_tiles_with_four_k_slices_where_the_tiles_are_few(bits):
    """End of round 6: at most 64 narrow 64 x 64 tiles over K >= 4096 take four K slices by default (1024 x 4096: 16 column tiles x 2 row tiles x 4 = 128 blocks,
    combined by ticket through the slabs)."""
    lin = _make_layer(1024, 4096, bits, 128, torch.float16, seed=97 + bits)
    for M in (96, 128, 250):
        x = torch.from_numpy(O.gen_x(M, 4096, seed=M)).to(DEV)
        name = _kernel_name(lin, x)
        assert name == f"gemm_w{bits}_mma_kernel<64x64>", name
        y = lin(x)
        torch.cuda.synchronize()
        _compare(f"mma_narrow/four_slices/w{bits}/M{M}", y, _oracle_from_layer(lin, x), lin.output_dtype.value, abs_gate=1e-3)


@pytest.mark.parametrize("K", [64, 128, 256, 384])
def test_mma_kernel_fewer_k_steps_than_stages(K):
    """One to six K steps: fewer than the LDS stages / the register ring of every tile variant (the run-ahead requests repeat the
    last step and are never consumed), narrow and wide tiles, with and without K slices."""
    from gemlite_amd.core import _hip_matmul
    lin = _make_layer(256, K, 4, 64, torch.bfloat16, seed=90 + K)
    for M in (40, 200):
        x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(torch.bfloat16).to(DEV)
        y_or = _oracle_from_layer(lin, x)
        for tuning in ((0, 1, 1, 0), (0, 1, 2, 0), (0, 1, 4, 0), (0, 1, 8, 0), (0, 2, 4, 0), (0, 1, 20, 0), (0, 1, 24, 0), (0, 2, 24, 0)):
            try:
                name = _kernel_name(lin, x, 4, tuning)
            except Exception:
                continue  # this K does not divide the variant's step (256-k steps of the small tiles) or has fewer steps than slices
            if "mma_kernel" not in name:
                continue
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, tuning)
            torch.cuda.synchronize()
            _compare(f"mma_short_k/K{K}/M{M}/{tuning}", y, y_or, lin.output_dtype.value, abs_gate=1e-3, extra=dict(kernel=name))


def test_mma_kernel_wide_tiles_prefill_shape_and_modes():
    """What the planner picks for prefill (256 x 256 tiles when they alone fill the chip) at 2048 x 4096 x 8192, every
    W_group_mode, checked on column blocks against the oracle and on all outputs against the 128-column tiles."""
    from gemlite_amd.core import _hip_matmul
    N, K, M = 8192, 4096, 2048
    for mode, kw in (("fma", {}), ("sub_mul", dict(fma=False)), ("symmetric", dict(zeros_kind="none")), ("int_zero", dict(zeros_kind="int"))):
        lin = _make_layer(N, K, 4, 128, torch.bfloat16, seed=81, **kw)
        x = torch.from_numpy(O.gen_x(M, K, seed=3).astype(np.float32)).to(torch.bfloat16).to(DEV)
        name = _kernel_name(lin, x)
        assert name == "gemm_w4_mma_kernel<256x256>", name
        y = lin(x)
        torch.cuda.synchronize()
        for c0 in (0, N - 256):
            cols = slice(c0, c0 + 256)
            _compare(f"mma_wide/prefill/{mode}/cols{c0}", y[:64, cols], _oracle_columns(lin, x[:64], cols), lin.output_dtype.value,
                     extra=dict(kernel=name))
        y2 = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, (0, 0, 8, 0))
        torch.cuda.synchronize()
        d = (y.float() - y2.float()).abs()  # two summation orders, each rounded to bf16 once: a few ulps at most
        ref = y2.float().abs()
        assert bool((d <= 0.02 * ref + 0.02 * float(ref.mean())).all()), (mode, float(d.max()))
        assert float(d.mean()) < 0.004 * float(ref.mean()), (mode, float(d.mean()))


@pytest.mark.parametrize("gs", [128, 64])
@pytest.mark.parametrize("N,K", [(1024, 11008), (1536, 8960), (1024, 896)])
def test_llm_shapes_with_odd_k_never_hit_the_coverage_kernel(N, K, gs):
    """K = 11008 (Llama-2-7B down_proj) / 8960 (Qwen2.5-1.5B) / 896 (Qwen2.5-0.5B: 128 * 7): shapes the power-of-two
    chunking of the round-1 kernels rejected (they fell to generic_matmul_kernel)."""
    lin = _make_layer(N, K, 4, gs, torch.float16, seed=50)
    for M in (1, 5, 32, 64, 200):
        x = torch.from_numpy(O.gen_x(M, K, seed=M)).to(DEV)
        name = _kernel_name(lin, x)
        assert not name.startswith("generic"), (M, name)
        y = lin(x)
        torch.cuda.synchronize()
        _compare(f"oddk/{N}x{K}/g{gs}/M{M}", y, _oracle_from_layer(lin, x), 1, extra=dict(kernel=name))


def _oracle_columns(lin, x, cols, scales_x=None):
    """Oracle on a subset of output columns (packed layers): the float64 evaluation of a 16384 x 16384 layer at M = 256
    is ~140 GFLOP on the host, so full-size tests check column blocks against the oracle and ALL outputs against a
    second, independently written kernel family."""
    meta = lin.get_meta_args()
    s = O.to_f64(lin.scales.data)[..., cols] if lin.scales.numel() else None
    z = O.to_f64(lin.zeros.data)[..., cols] if lin.zeros.numel() > 1 else (O.to_f64(lin.zeros.data).reshape(-1) if lin.zeros.numel() == 1 else None)
    Wp = lin.W_q.data[:, cols].cpu().numpy()
    return O.forward_packed(O.to_f64(x).reshape(-1, x.shape[-1]), Wp, s, z, W_nbits=lin.W_nbits, group_size=lin.group_size,
                            W_group_mode=meta[10], channel_scale_mode=meta[9], scales_x=scales_x,
                            zero_is_scalar=lin.zeros.numel() == 1, pack_bits=32)


def test_config5_a16w2_16384_m256():
    """BASELINE config 5, prefill half: A16W2 g128 16384 x 16384, M = 256 (32-bit buffer offsets at 2^28 words)."""
    from gemlite_amd.core import _hip_matmul
    N = K = 16384
    lin = _make_layer(N, K, 2, 128, torch.float16, seed=61)
    x = torch.from_numpy(O.gen_x(256, K, seed=9)).to(DEV)
    name = _kernel_name(lin, x)
    assert name.startswith("gemm_w2_mma_kernel<256x128>"), name
    y = lin(x)
    torch.cuda.synchronize()
    for c0 in (0, 8192 - 128, N - 256):  # first / middle / last column blocks
        cols = slice(c0, c0 + 256)
        _compare(f"cfg5/a16w2/M256/cols{c0}", y[:, cols], _oracle_columns(lin, x, cols), 1, extra=dict(kernel=name))
    # every output against the LDS-staged streaming kernel (different code, same inputs)
    y2 = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, (1, 0, 0, 0))
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs()
    assert float(d.max()) < 0.02 * float(y2.float().abs().mean()) + 1e-3, float(d.max())


@pytest.mark.parametrize("M", [1, 256])
def test_config5_fp8_fp8_16384(M):
    """BASELINE config 5: FP8 x FP8 (per-token activation scales x per-channel weight scales), 16384 x 16384."""
    from gemlite_amd.core import _hip_matmul
    N = K = 16384
    g = torch.Generator().manual_seed(70 + M)
    W = (torch.randn(N, K, generator=g) / 30).half()
    lin = gemlite_amd.helper.A8W8_fp8_dynamic(device=DEV, dtype=torch.float16).from_weights(W)
    del W
    x = (torch.randn(M, K, generator=g) / 10).half().to(DEV)
    name = _kernel_name(lin, x)
    y = lin(x)
    torch.cuda.synchronize()
    xq, sx = O.scale_activations_per_token(x, O.FP8E4)
    sw = O.to_f64(lin.scales.data).reshape(-1)
    for c0 in (0, 8192 - 64, N - 128):
        cols = slice(c0, c0 + 128)
        y_or = (xq @ O.to_f64(lin.W_q.data[:, cols])) * (sx.astype(np.float64) * sw[cols].reshape(1, -1))
        _compare(f"cfg5/fp8/M{M}/cols{c0}", y[:, cols], y_or, 1, abs_gate=5e-3, extra=dict(kernel=name))
    # all outputs: MFMA kernel vs the one-wave-per-column streaming kernel on the same quantised activations
    xq_g, sx_g = scale_activations_per_token(x, torch.float8_e4m3fn)
    y2 = _hip_matmul(xq_g, lin.W_q, lin.scales, lin.zeros, sx_g, lin.get_meta_args(), -1, (1, 0, 0, 0))
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs()
    assert float(d.max()) < 0.02 * float(y2.float().abs().mean()) + 1e-3, (name, float(d.max()))


def test_forward_functional_custom_op_matches_the_module_and_compiles():
    """gemlite::forward_functional — the entry point hqq / vLLM use (reference core.py:128-206): direct call, opcheck
    (schema, fake impl, dispatch), and a torch.compile(fullgraph=True) module forward against eager."""
    lin = _make_layer(1024, 2048, 4, 128, torch.float16, seed=77)
    bias = (torch.randn(1024) / 10).half().to(DEV)
    lin.bias = torch.nn.Parameter(bias, requires_grad=False)
    op = torch.ops.gemlite.forward_functional
    for shape in ((1, 2048), (3, 5, 2048)):
        x = (torch.randn(*shape) / 10).half().to(DEV)
        y_mod = lin(x)
        y_fn = gemlite_amd.core.forward_functional(x, lin.bias, lin.get_tensor_args(), lin.get_meta_args(), -1)
        y_op = op(x, lin.bias, lin.get_tensor_args(), lin.get_meta_args(), -1)
        assert y_fn.shape == x.shape[:-1] + (1024,) and torch.equal(y_mod, y_fn) and torch.equal(y_mod, y_op)
        for mt in ("GEMV", "GEMV_REVSPLITK", "GEMM_SPLITK", "GEMM"):
            y_m = op(x, None, lin.get_tensor_args(), lin.get_meta_args(), gemlite_amd.core.GEMLITE_MATMUL_TYPES_MAPPING[mt])
            _compare(f"functional/{mt}/{shape}", y_m.reshape(-1, 1024), _oracle_from_layer(lin, x), 1)
    x = (torch.randn(4, 2048) / 10).half().to(DEV)
    torch.library.opcheck(op, (x, lin.bias, lin.get_tensor_args(), lin.get_meta_args(), -1),
                          test_utils=("test_schema", "test_faketensor"))

    class Block(torch.nn.Module):
        def __init__(self, layer):
            super().__init__()
            self.layer = layer

        def forward(self, t):
            return torch.nn.functional.silu(self.layer(t)) * 2.0

    blk = Block(lin)
    y_eager = blk(x)
    y_comp = torch.compile(blk, fullgraph=True)(x)
    torch.cuda.synchronize()
    assert y_comp.shape == y_eager.shape
    assert float((y_comp.float() - y_eager.float()).abs().max()) < 2e-3


@pytest.mark.parametrize("kind", ["int8", "fp8"])
def test_fused_activation_quant_at_m1_is_bit_identical_to_the_two_launch_path(kind):
    """at M = 1 the per-token activation quantisation runs inside the matmul kernel's prologue.
    Same arithmetic as scale_activations_per_token + the streaming matmul -> bit-identical outputs."""
    torch.manual_seed(11)
    W = (torch.randn(4096, 4096) / 30).half()
    proc = (gemlite_amd.helper.A8W8_int8_dynamic if kind == "int8" else gemlite_amd.helper.A8W8_fp8_dynamic)(device=DEV, dtype=torch.float16)
    lin = proc.from_weights(W)
    for shape in ((1, 4096), (1, 1, 4096), (4096,)):
        x = (torch.randn(*shape) / 10).half().to(DEV)
        y_fused = lin(x)
        gemlite_amd.core.FUSE_ACT_QUANT_M1 = False
        try:
            y_two = lin(x)
        finally:
            gemlite_amd.core.FUSE_ACT_QUANT_M1 = True
        torch.cuda.synchronize()
        assert y_fused.shape == y_two.shape and torch.equal(y_fused, y_two), (kind, shape)
    x = (torch.randn(1, 4096) / 10).half().to(DEV)
    xq, sx = O.scale_activations_per_token(x, O.INT8 if kind == "int8" else O.FP8E4)
    y_or = (xq @ O.to_f64(lin.W_q.data)) * (sx.astype(np.float64) * O.to_f64(lin.scales.data).reshape(1, -1))
    _compare(f"fused-quant/{kind}", lin(x), y_or, 1, abs_gate=5e-3)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("proc", ["A8W4_fp8", "A8W2_fp8", "A8W158_int8"])
def test_packed_dynamic_layers_quantise_the_row_inside_the_decode_kernel(proc, tdt):
    """M = 1 of A8W4 / A8W2 fp8-dynamic and BitNet int8-dynamic (helper.py:502-615, 1006-1062): `layer(x)` is ONE launch — the decode kernel
    requests its weights, then quantises the 16-bit row per token into LDS (gemv_a8wn_kernel<..., FQ>).  Bit-identical to quantiser +
    the same kernel on the quantised row (same arithmetic, same summation order), for grouped and channel-wise metadata, 3-d inputs, and
    launch after launch with different rows."""
    from gemlite_amd import core
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    torch.manual_seed(43)
    N, K = 1024, 4096
    if proc == "A8W158_int8":
        lins = [H.A8W158_INT_dynamic(device=DEV, dtype=tdt).from_weights(torch.randint(-1, 2, (N, K)).to(tdt), torch.tensor(0.02))]
        qdt = torch.int8
    else:
        nbits = 4 if proc == "A8W4_fp8" else 2
        lins = []
        for gs, post in ((128, False), (K, True)):
            W_q, sc, zr = O.gen_data(N, K, nbits, gs, seed=50 + nbits)
            lins.append(H.A8Wn_HQQ_INT_dynamic(device=DEV, dtype=tdt, post_scale=post, W_nbits=nbits).from_weights(
                torch.from_numpy(W_q), torch.from_numpy(sc).to(tdt), torch.from_numpy(zr).to(tdt)))
        qdt = torch.float8_e4m3fn
    for lin in lins:
        for rep, shape in enumerate(((1, K), (1, 1, K), (K,))):
            x = (torch.randn(*shape, device=DEV) * (0.05 + 0.1 * rep)).to(tdt)
            a = core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
            a.matmul_type, a.M, a.x, a.out = -1, 1, x.data_ptr(), 0x1000
            a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = K, 1, N, 1
            a.input_dtype = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
            name = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
            assert "fused_quant_kernel" in name, name
            y_fused = lin(x)
            xq, sx = scale_activations_per_token(x.reshape(1, K), w_dtype=qdt)
            y_two = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1)
            torch.cuda.synchronize()
            assert torch.equal(y_fused.reshape(1, N), y_two), (proc, tdt, shape, float((y_fused.reshape(1, N).float() - y_two.float()).abs().max()))
        core.FUSE_ACT_QUANT_M1 = False
        try:
            assert torch.equal(lin(x), y_fused)  # the switch: quantiser + matmul
        finally:
            core.FUSE_ACT_QUANT_M1 = True


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("proc", ["A16W8_INT8", "A16W8_INT8_post", "A16W8_FP8"])
def test_a16w8_rows_kernel_against_the_oracle_and_the_streaming_kernel(proc, tdt):
    """8-bit weight-only layers (helper.py:88-171) on a16w8_rows_kernel (round 4): int8 / fp8 weights converted in registers (int8 -> fp16
    through the 1024 + (b + 128) bit pattern, exact), two v_mfma_f32_16x16x32 per 64-k chunk and 16 rows, pre- and post-scale, every
    row-tile height, ragged M, 64-row tiles along grid.y above 64 rows, K an odd multiple of 64; against the float64 evaluation of the
    stored tensors and against kmajor_w8a16_kernel (tuning[0] = 7)."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    torch.manual_seed(41)
    N, K = 1024, 2048 + 64
    W = (torch.randn(N, K) / 30).to(tdt)
    W[3, :] *= 6.0  # a column whose int8 codes use the whole range
    mk = {"A16W8_INT8": lambda: H.A16W8(device=DEV, dtype=tdt), "A16W8_INT8_post": lambda: H.A16W8(device=DEV, dtype=tdt, post_scale=True),
          "A16W8_FP8": lambda: H.A16W8_FP8(device=DEV, dtype=tdt)}[proc]
    lin = mk().from_weights(W)
    out_code = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
    # one row, K % 1024 == 0: a16w8_decode_kernel (a wave per column, weights first, x through LDS) — against the oracle and the rows kernel
    for K1 in (2048, 9216):
        W1 = (torch.randn(512, K1) / 30).to(tdt)
        lin1 = mk().from_weights(W1)
        x1 = (torch.randn(1, K1, device=DEV) / 10).to(tdt)
        assert _kernel_name(lin1, x1) == "a16w8_decode_kernel<tile16,16w>", _kernel_name(lin1, x1)
        y1 = lin1(x1)
        y1_rows = _hip_matmul(x1, lin1.W_q, lin1.scales, lin1.zeros, None, lin1.get_meta_args(), -1, (4, 0, 0, 0))
        torch.cuda.synchronize()
        _compare(f"a16w8-decode/{proc}/{str(tdt)[6:]}/K{K1}", y1, _oracle_from_layer(lin1, x1), out_code, abs_gate=5e-3)
        assert float((y1.float() - y1_rows.float()).abs().mean() / y1_rows.float().abs().mean()) < (2e-3 if tdt == torch.float16 else 8e-3)
    for M in (1, 2, 16, 17, 33, 64, 65, 200):
        x = (torch.randn(M, K, device=DEV) / 10).to(tdt)
        name = _kernel_name(lin, x)
        assert name == "a16w8_rows_kernel<%s>" % ("16x16" if M <= 16 else ("32x16" if M <= 32 else "64x16")), (M, name)
        y = lin(x)
        y_old = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (7, 0, 0, 0))
        torch.cuda.synchronize()
        y_or = _oracle_from_layer(lin, x)
        _compare(f"a16w8-rows/{proc}/{str(tdt)[6:]}/M{M}", y, y_or, out_code, abs_gate=5e-3)
        rel = float((y.float() - y_old.float()).abs().mean() / y_old.float().abs().mean())
        assert rel < (2e-3 if tdt == torch.float16 else 8e-3), (proc, M, rel)


ROUND4_W8_ROWS, W8_ROWS_LDS_FROM_1 = 524288, 1048576  # tuning[3]: keep the register-fed rows kernels / take the x-through-LDS one below 17 rows too


def _w8_lds_name(fam, M):
    return "%s_rows_lds_kernel<%dx16>" % (fam, 16 if M <= 16 else (32 if M <= 32 else (48 if M <= 48 else 64)))


@pytest.mark.parametrize("kind", ["int8", "fp8e4", "fp8e5"])
def test_a8w8_rows_lds_kernel(kind):
    """Round 6: A8W8 at 1 .. 64 rows on w8_rows_lds_kernel (gemm_w8_rows.hip; replaces gemm_splitK_kernels.py:277-450 for BASELINE config 4's
    decode batches): x through LDS in whole cache lines (every row-tile height = every piece geometry), weights in 256-k chunks dealt over
    8 waves — K = 256 (seven waves idle), 256 * 9 (uneven deal), 16384 (eight chunks per wave), ragged M; against the oracle on the kernel's
    own quantised inputs, and for int8 BIT-EXACT against the register-fed kernel of round 3."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    code = {"int8": O.INT8, "fp8e4": O.FP8E4, "fp8e5": O.FP8E5}[kind]
    qdt = {"int8": torch.int8, "fp8e4": torch.float8_e4m3fn, "fp8e5": torch.float8_e5m2}[kind]
    ran = set()
    for (N, K) in ((512, 4096), (48, 256 * 9), (256, 16384), (64, 256), (32, 768)):
        g = torch.Generator().manual_seed(N + K)
        W = (torch.randn(N, K, generator=g) / 30).half()
        proc = H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16) if kind == "int8" else \
            H.A8W8_dynamic(device=DEV, dtype=torch.float16, fp8=qdt)
        lin = proc.from_weights(W)
        for M in (1, 2, 7, 16, 17, 29, 32, 33, 48, 50, 64):
            x = (torch.randn(M, K, generator=g) / 10).half().to(DEV)
            xq, sx = scale_activations_per_token(x, qdt)
            tun = (4, 0, 0, 0)  # (tuning[0] = 4: one row too, and past the x re-read budget)
            name = _kernel_name(lin, xq, -1, tun)
            assert name == _w8_lds_name("a8w8", M), (M, name)
            ran.add(name)
            y = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, tun)
            y_r4 = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, (4, 0, 0, ROUND4_W8_ROWS))
            torch.cuda.synchronize()
            assert _kernel_name(lin, xq, -1, (4, 0, 0, ROUND4_W8_ROWS)).startswith("a8w8_rows_kernel<")
            xq_o, sx_o = O.scale_activations_per_token(x, code)
            y_or = (xq_o @ O.to_f64(lin.W_q.data)) * (sx_o.astype(np.float64) * O.to_f64(lin.scales.data).reshape(1, -1))
            _compare(f"a8w8_rows_lds/{kind}/{N}x{K}/M{M}", y, y_or, 1, abs_gate=5e-3, extra=dict(kernel=name))
            if kind == "int8":
                assert torch.equal(y, y_r4), (N, K, M)
            else:
                assert float((y.float() - y_r4.float()).abs().mean() / y_r4.float().abs().mean()) < 1e-3, (N, K, M)
    assert len(ran) == 4, ran
    # the default from 2 rows (no tuning) on a layer whose 16-column blocks are one resident round
    lin = H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16).from_weights((torch.randn(4096, 4096) / 30).half())
    for M, want in ((2, "a8w8_rows_lds_kernel<16x16>"), (17, "a8w8_rows_lds_kernel<32x16>"), (40, "a8w8_rows_lds_kernel<48x16>"), (64, "a8w8_rows_lds_kernel<64x16>")):
        assert _kernel_name(lin, torch.empty(M, 4096, dtype=torch.int8)) == want, (M, _kernel_name(lin, torch.empty(M, 4096, dtype=torch.int8)))
    # more blocks than CUs: the 64-KB form (two blocks per CU), 4 .. 16 rows by default, up to 32 when forced — bit-exact against the round-3 kernel
    g = torch.Generator().manual_seed(5)
    lin = H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16).from_weights((torch.randn(8192, 1024, generator=g) / 30).half())
    for M, tun, want in ((4, (0, 0, 0, 0), "a8w8_rows_lds_kernel<16x16,2/cu>"), (16, (0, 0, 0, 0), "a8w8_rows_lds_kernel<16x16,2/cu>"), (27, (4, 0, 0, 0), "a8w8_rows_lds_kernel<32x16,2/cu>")):
        x = (torch.randn(M, 1024, generator=g) / 10).half().to(DEV)
        xq, sx = scale_activations_per_token(x, torch.int8)
        assert _kernel_name(lin, xq, -1, tun) == want, (M, _kernel_name(lin, xq, -1, tun))
        y = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, tun)
        y_r4 = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, (4, 0, 0, ROUND4_W8_ROWS))
        torch.cuda.synchronize()
        assert torch.equal(y, y_r4), M


def test_w8_rows_lds_kernel_reads_rows_by_their_stride():
    """w8_rows_lds_kernel builds the source addresses of its LDS-DMA pieces from stride_xm (rows of a wider tensor, e.g. one projection's slice of a
    fused activation buffer): a strided x must give the bits of the contiguous copy, for every row-tile height, 8-bit and 16-bit x."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    g = torch.Generator().manual_seed(77)
    N, K = 256, 1024
    W = (torch.randn(N, K, generator=g) / 30).half()
    lin8 = H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16).from_weights(W)
    lin16 = H.A16W8(device=DEV, dtype=torch.float16).from_weights(W)
    for M in (5, 20, 40, 64):
        wide = (torch.randn(M, 3 * K, generator=g) / 10).half().to(DEV)
        x = wide[:, K:2 * K]                              # row stride 3 K, 16-byte aligned
        assert not x.is_contiguous() and x.stride(0) == 3 * K
        tun = (4, 0, 0, 0)
        assert _kernel_name(lin16, x, -1, tun).startswith("a16w8_rows_lds_kernel<")
        y = _hip_matmul(x, lin16.W_q, lin16.scales, lin16.zeros, None, lin16.get_meta_args(), -1, tun)
        y_c = _hip_matmul(x.contiguous(), lin16.W_q, lin16.scales, lin16.zeros, None, lin16.get_meta_args(), -1, tun)
        xq, sx = scale_activations_per_token(x.contiguous(), torch.int8)
        wide_q = torch.zeros(M, 3 * K, dtype=torch.int8, device=DEV)
        wide_q[:, K:2 * K] = xq
        xq_s = wide_q[:, K:2 * K]
        assert _kernel_name(lin8, xq_s, -1, tun).startswith("a8w8_rows_lds_kernel<")
        z = _hip_matmul(xq_s, lin8.W_q, lin8.scales, lin8.zeros, sx, lin8.get_meta_args(), -1, tun)
        z_c = _hip_matmul(xq, lin8.W_q, lin8.scales, lin8.zeros, sx, lin8.get_meta_args(), -1, tun)
        torch.cuda.synchronize()
        assert torch.equal(y, y_c) and torch.equal(z, z_c), M


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("proc", ["A16W8_INT8", "A16W8_INT8_post", "A16W8_FP8", "A16W8_FP8e5"])
def test_a16w8_rows_lds_kernel(proc, tdt):
    """Round 6: 8-bit weight-only layers (helper.py:88-171) at 1 .. 64 rows (and 64-row blocks along grid.y above) on w8_rows_lds_kernel: 16-bit x
    through LDS (the lane's sixteen k = two adjacent 16-byte slots: the permuted-k A fragments), int8 / e4m3 / e5m2 weights converted in
    registers exactly as a16w8_rows_kernel does (gl_w8cvt.h), pre- and post-scale; against the float64 oracle and the round-4 kernel."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    torch.manual_seed(47)
    out_code = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
    ran = set()
    for N, K, Ms in ((1024, 2048 + 256, (1, 2, 16, 17, 33, 48, 49, 64, 65, 200)), (64, 256, (5, 20, 40, 64)), (48, 768, (3, 31, 47, 63)), (128, 8192, (16, 32, 48, 64))):
        W = (torch.randn(N, K) / 30).to(tdt)
        W[3, :] *= 6.0  # a column whose int8 codes use the whole range
        if proc == "A16W8_FP8e5":
            sc = (W.float().abs().amax(dim=1, keepdim=True) / 57344.0).clamp_(min=1e-6)
            lin = H.A16W8(device=DEV, dtype=tdt).from_weights((W.float() / sc).to(torch.float8_e5m2), scales=sc)
        else:
            mk = {"A16W8_INT8": lambda: H.A16W8(device=DEV, dtype=tdt), "A16W8_INT8_post": lambda: H.A16W8(device=DEV, dtype=tdt, post_scale=True),
                  "A16W8_FP8": lambda: H.A16W8_FP8(device=DEV, dtype=tdt)}[proc]
            lin = mk().from_weights(W)
        for M in Ms:
            x = (torch.randn(M, K, device=DEV) / 10).to(tdt)
            tun = (4, 0, 0, W8_ROWS_LDS_FROM_1 if M < 4 else 0)   # (by default from 4 rows)
            name = _kernel_name(lin, x, -1, tun)
            assert name == _w8_lds_name("a16w8", M), (M, name)
            ran.add(name)
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tun)
            y_r4 = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (4, 0, 0, ROUND4_W8_ROWS))
            torch.cuda.synchronize()
            assert _kernel_name(lin, x, -1, (4, 0, 0, ROUND4_W8_ROWS)).startswith("a16w8_rows_kernel<")
            _compare(f"a16w8-rows-lds/{proc}/{str(tdt)[6:]}/{N}x{K}/M{M}", y, _oracle_from_layer(lin, x), out_code, abs_gate=5e-3)
            rel = float((y.float() - y_r4.float()).abs().mean() / y_r4.float().abs().mean())
            assert rel < (2e-3 if tdt == torch.float16 else 8e-3), (proc, M, rel)
    assert len(ran) == 4, ran
    # more blocks than CUs: the 64-KB form (two blocks per CU; 256- / 128-byte row pieces)
    N, K = 8192, 1024
    W = (torch.randn(N, K) / 30).to(tdt)
    if proc == "A16W8_FP8e5":
        sc = (W.float().abs().amax(dim=1, keepdim=True) / 57344.0).clamp_(min=1e-6)
        lin = H.A16W8(device=DEV, dtype=tdt).from_weights((W.float() / sc).to(torch.float8_e5m2), scales=sc)
    else:
        lin = mk().from_weights(W)
    for M, tun, want in ((4, (0, 0, 0, 0), "a16w8_rows_lds_kernel<16x16,2/cu>"), (16, (0, 0, 0, 0), "a16w8_rows_lds_kernel<16x16,2/cu>"), (29, (4, 0, 0, 0), "a16w8_rows_lds_kernel<32x16,2/cu>")):
        x = (torch.randn(M, K, device=DEV) / 10).to(tdt)
        assert _kernel_name(lin, x, -1, tun) == want, (M, _kernel_name(lin, x, -1, tun))
        y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tun)
        torch.cuda.synchronize()
        _compare(f"a16w8-rows-lds-2cu/{proc}/{str(tdt)[6:]}/M{M}", y, _oracle_from_layer(lin, x), out_code, abs_gate=5e-3)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("proc", ["A16W8_INT8", "A16W8_INT8_post", "A16W8_FP8", "A16W8_FP8e5"])
def test_a16w8_tile_kernel_above_64_rows(proc, tdt):
    """8-bit weight-only layers above 64 rows (round 4): the 8-wave MFMA tile kernel with the K-contiguous 8-bit geometries
    (Geo<KW8I / KW8F / KW8B>: int8 / e4m3 / e5m2 converted in registers at scale 1, the channel scale in the epilogue — pre-scale layers
    included).  Every tile height, split-K and unsplit, ragged M; against the float64 oracle and the rows kernel (tuning[0] = 4)."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    torch.manual_seed(43)
    out_code = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
    for N, K, Ms in ((1024, 2048, (65, 130, 300)), (256, 4096 + 128, (100,)), (4096, 1024, (513,))):
        W = (torch.randn(N, K) / 30).to(tdt)
        W[3, :] *= 6.0
        if proc == "A16W8_FP8e5":
            sc = (W.float().abs().amax(dim=1, keepdim=True) / 57344.0).clamp_(min=1e-6)
            lin = H.A16W8(device=DEV, dtype=tdt).from_weights((W.float() / sc).to(torch.float8_e5m2), scales=sc)
        else:
            mk = {"A16W8_INT8": lambda: H.A16W8(device=DEV, dtype=tdt), "A16W8_INT8_post": lambda: H.A16W8(device=DEV, dtype=tdt, post_scale=True),
                  "A16W8_FP8": lambda: H.A16W8_FP8(device=DEV, dtype=tdt)}[proc]
            lin = mk().from_weights(W)
        for M in Ms:
            x = (torch.randn(M, K, device=DEV) / 10).to(tdt)
            name = _kernel_name(lin, x)
            assert name.startswith("gemm_a16w8_kernel<"), (proc, M, name)
            y = lin(x)
            y_rows = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (4, 0, 0, 0))
            torch.cuda.synchronize()
            _compare(f"a16w8-tile/{proc}/{str(tdt)[6:]}/{N}x{K}/M{M}", y, _oracle_from_layer(lin, x), out_code, abs_gate=5e-3)
            rel = float((y.float() - y_rows.float()).abs().mean() / y_rows.float().abs().mean())
            assert rel < (2e-3 if tdt == torch.float16 else 8e-3), (proc, M, rel)
            for tun in ((2, 1, 1, 0), (2, 2, 2, 0), (2, 1, 4, 0), (2, 4, 8, 0), (2, 1, 32, 0), (2, 2, 32, 0)):  # (tile kernel, split-K, 32-row units per tile; 32 = the narrow 64 x 64 tiles)
                if tun[2] == 32 and K % 256 != 0:
                    continue
                assert ("<64x64>" in _kernel_name(lin, x, tuning=tun)) == (tun[2] == 32), tun
                y_t = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tun)
                torch.cuda.synchronize()
                rel = float((y_t.float() - y_rows.float()).abs().mean() / y_rows.float().abs().mean())
                assert rel < (2e-3 if tdt == torch.float16 else 8e-3), (proc, M, tun, rel)


@pytest.mark.parametrize("proc", ["A8W4", "A8W2", "A8W158"])
def test_narrow_tiles_with_8bit_activations(proc):
    """The narrow 64 x 64 tiles (KH = 4) of the 8-wave kernel for fp8 / int8 activations x packed words (round 4, late; forced with
    tuning[2] = 32, K slices in tuning[1]): against the 128-column tiles of the same kernel — BitNet int8 bit for bit (exact int32
    accumulation), fp8 within the fp32 summation-order tolerance — and, through them, against everything those are pinned to."""
    from gemlite_amd import core
    H = gemlite_amd.helper
    torch.manual_seed(5)
    N, K = 512, 2048 + 256
    if proc == "A8W158":
        lin = H.A8W158_INT_dynamic(device=DEV).from_weights(torch.randint(-1, 2, (N, K)).half(), torch.tensor(0.02))
    else:
        nb = 4 if proc == "A8W4" else 2
        W_q = torch.randint(0, 2 ** nb, (N, K), dtype=torch.int32).to(torch.uint8).to(DEV)
        s = (torch.rand(N * K // 128, 1, device=DEV) * 0.01 + 0.001).half()
        z = (torch.rand(N * K // 128, 1, device=DEV) * (2 ** nb - 1)).half()
        lin = (H.A8W4_HQQ_INT_dynamic if nb == 4 else H.A8W2_HQQ_INT_dynamic)(device=DEV, dtype=torch.float16).from_weights(W_q, s, z)
    for M in (65, 200):
        x = (torch.randn(M, K) / 4).half().to(DEV)
        try:
            core.TUNING_OVERRIDE = (0, 0, 2, 0)  # 64 x 128 tiles
            assert "x128>" in _kernel_name(lin, x, tuning=(0, 0, 2, 0))
            y0 = lin(x)
            for tun in ((0, 1, 32, 0), (0, 3, 32, 0)):
                core.TUNING_OVERRIDE = tun
                name = _kernel_name(lin, x, tuning=tun)
                assert name == ("gemm_a8w4_mma_kernel<64x64>" if proc == "A8W4" else "gemm_a8w2_mma_kernel<64x64>"), (tun, name)
                y = lin(x)
                torch.cuda.synchronize()
                if proc == "A8W158":
                    assert torch.equal(y, y0), (M, tun)
                else:
                    rel = float((y.float() - y0.float()).abs().mean() / y0.float().abs().mean())
                    assert rel < 2e-3, (proc, M, tun, rel)
        finally:
            core.TUNING_OVERRIDE = None


@pytest.mark.parametrize("kind", ["int8", "fp8"])
def test_a8w8_decode_kernel_against_the_round2_kernels_and_the_oracle(kind):
    """a8w8_decode_kernel (round 4: one wave per column, the weight row requested before anything else; FUSED form quantises x under that
    round trip): K below / at / above one batch of pieces (2048 / 8192 / 16384), fp16 and bf16 activations — int8 bit-identical to the
    round-2 kernels (tuning[0] = 7) and to the two-launch path, fp8 within the fp32 summation-order tolerance; both against the oracle."""
    from gemlite_amd import core
    from gemlite_amd.core import _hip_matmul
    torch.manual_seed(37)
    qdt = torch.int8 if kind == "int8" else torch.float8_e4m3fn
    for N, K, tdt in ((1024, 2048, torch.float16), (512, 8192, torch.bfloat16), (256, 16384, torch.float16), (8192, 1024, torch.float16)):
        W = (torch.randn(N, K) / 30).to(tdt)
        lin = (gemlite_amd.helper.A8W8_int8_dynamic if kind == "int8" else gemlite_amd.helper.A8W8_fp8_dynamic)(device=DEV, dtype=tdt).from_weights(W)
        x = (torch.randn(1, K, device=DEV) / 10).to(tdt)
        xq, sx = scale_activations_per_token(x, w_dtype=qdt)
        a = core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
        a.matmul_type, a.M, a.x, a.out = -1, 1, x.data_ptr(), 0x1000
        a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = K, 1, N, 1
        a.input_dtype = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
        new_default = kind == "int8" or N <= 4096  # fp8 above one round of tiles keeps the round-2 kernels (measured faster there)
        assert _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode() == ("a8w8_decode_fused_quant_kernel<tile16,16w>" if new_default else "kmajor_fused_quant_kernel")
        if not new_default:
            continue
        y_fused = lin(x)                                                                               # new fused kernel
        y_fused_old = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (7, 0, 0, 0))   # round-2 fused kernel
        y_two = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1)             # quantiser + new decode kernel
        y_two_old = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1, (7, 0, 0, 0))
        torch.cuda.synchronize()
        assert torch.equal(y_fused, y_two), (kind, N, K)
        if kind == "int8":
            assert torch.equal(y_fused, y_fused_old) and torch.equal(y_two, y_two_old), (kind, N, K)
        else:
            for other in (y_fused_old, y_two_old):
                rel = float((y_fused.float() - other.float()).abs().mean() / other.float().abs().mean())
                assert rel < 2e-3, (kind, N, K, rel)
        code = O.INT8 if kind == "int8" else O.FP8E4
        xq_o, sx_o = O.scale_activations_per_token(x, code)
        y_or = (xq_o @ O.to_f64(lin.W_q.data)) * (sx_o.astype(np.float64) * O.to_f64(lin.scales.data).reshape(1, -1))
        _compare(f"a8w8-decode/{kind}/{N}x{K}", y_fused, y_or, gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value, abs_gate=5e-3)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("kind", ["int8", "fp8"])
def test_cooperative_activation_quant_from_two_rows_is_bit_identical_to_the_two_launch_path(kind, tdt):
    """SURVEY.md §8 f1 / VERDICT r3 #5: for 2 .. 64 rows the A8W8 launch carries producer blocks that quantise the rows of x into the
    workspace (csrc/gl_coopquant.h) while the tile blocks stream their weights and wait for the row flags — `layer(x)` is ONE launch.
    Same arithmetic as scale_activations_per_token (quant_utils.py:231-253) + the same matmul kernel body: bit-identical outputs,
    every M class (16 / 32 / 64-row fragments), launch after launch with different activations in the same workspace (a stale cache
    line of the previous launch would show), flags left zero; above 64 rows the library says GEMLITE_ERR_NO_FUSED_QUANT and the host
    path runs quantiser + matmul."""
    from gemlite_amd import core
    from gemlite_amd.core import _hip_matmul
    torch.manual_seed(23)
    core.FUSE_ACT_QUANT_ROWS = True  # (opt-in: measured slower than two launches, see core.py)
    core._NO_FUSED_QUANT.clear()
    try:
        _cooperative_quant_cases(kind, tdt, core, _hip_matmul)
    finally:
        core.FUSE_ACT_QUANT_ROWS = False
    ws = _hip.workspace(torch.device(DEV), _hip.current_stream_handle(torch.device(DEV)), 0)
    torch.cuda.synchronize()
    assert int(ws[:4 * 1100].view(torch.int32).abs().sum()) == 0  # row flags and departure count are zero again


def _cooperative_quant_cases(kind, tdt, core, _hip_matmul):
    for N, K in ((1024, 2048), (4096, 4096)):
        W = (torch.randn(N, K) / 30).to(tdt)
        proc = (gemlite_amd.helper.A8W8_int8_dynamic if kind == "int8" else gemlite_amd.helper.A8W8_fp8_dynamic)(device=DEV, dtype=tdt)
        lin = proc.from_weights(W)
        qdt = torch.int8 if kind == "int8" else torch.float8_e4m3fn
        for M in (2, 5, 16, 17, 33, 64, 65, 256):
            want = "a8w8_rows_fq_kernel" if M <= 64 else "unsupported"
            for rep in range(3):
                x = (torch.randn(M, K, device=DEV) * (0.05 + 0.2 * rep)).to(tdt)
                x[min(rep, M - 1), 7] = 3.0 + rep  # an outlier row: its scale differs a lot from its neighbours'
                a = core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
                a.matmul_type, a.M, a.x, a.out = -1, M, x.data_ptr(), 0x1000
                a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = K, 1, N, 1
                a.input_dtype = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
                name = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
                assert name.startswith(want), (M, name)
                y_fused = lin(x)
                xq, sx = scale_activations_per_token(x, w_dtype=qdt)
                y_two = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1)
                torch.cuda.synchronize()
                assert torch.equal(y_fused, y_two), (kind, tdt, N, K, M, rep, float((y_fused.float() - y_two.float()).abs().max()))
            # 3-d input, and the switch
            x3 = (torch.randn(1, M, K, device=DEV) / 10).to(tdt)
            y3 = lin(x3)
            core.FUSE_ACT_QUANT_ROWS = False
            try:
                y3_two = lin(x3)
            finally:
                core.FUSE_ACT_QUANT_ROWS = True
            assert y3.shape == (1, M, N) and torch.equal(y3, y3_two)


@pytest.mark.parametrize("kind", ["int8", "fp8"])
def test_cooperative_activation_quant_when_no_block_produces_its_rows(kind):
    """The wait is bounded: a block that has polled long enough quantises the first missing row itself (identical bytes).  With the
    test bit tuning[3] & 32768 NO block quantises its own share, so every row arrives that way — the launch must still finish with the
    two-launch result.  (What happens when the owner of a row is not resident: CU masks, a co-running kernel, a grid above the chip.)"""
    from gemlite_amd.core import _hip_matmul
    torch.manual_seed(29)
    N, K = 1024, 2048
    W = (torch.randn(N, K) / 30).half()
    lin = (gemlite_amd.helper.A8W8_int8_dynamic if kind == "int8" else gemlite_amd.helper.A8W8_fp8_dynamic)(device=DEV, dtype=torch.float16).from_weights(W)
    qdt = torch.int8 if kind == "int8" else torch.float8_e4m3fn
    for M in (3, 40, 64):
        x = (torch.randn(M, K, device=DEV) / 8).half()
        y_steal = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (0, 0, 0, 32768))
        xq, sx = scale_activations_per_token(x, w_dtype=qdt)
        y_two = _hip_matmul(xq, lin.W_q, lin.scales, lin.zeros, sx, lin.get_meta_args(), -1)
        torch.cuda.synchronize()
        assert torch.equal(y_steal, y_two), (kind, M)
        y_again = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1)  # flags were left zero: the next in-launch quantisation is right as well
        torch.cuda.synchronize()
        assert torch.equal(y_again, y_two), (kind, M)


@pytest.mark.parametrize("M", [16, 48])
def test_cooperative_activation_quant_under_graph_replay(M):
    """One captured `layer(x)` launch replayed with x rewritten in place between replays: every replay starts from zeroed flags
    (the in-kernel reset) and reads the rows of ITS launch."""
    torch.manual_seed(31)
    N = K = 4096
    W = (torch.randn(N, K) / 30).half()
    lin = gemlite_amd.helper.A8W8_int8_dynamic(device=DEV, dtype=torch.float16).from_weights(W)
    x = (torch.randn(M, K, device=DEV) / 10).half()
    gemlite_amd.core.FUSE_ACT_QUANT_ROWS = True
    gemlite_amd.core._NO_FUSED_QUANT.clear()
    lin(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        lin(x)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = lin(x)
    torch.cuda.current_stream().wait_stream(s)
    gemlite_amd.core.FUSE_ACT_QUANT_ROWS = False
    try:
        for rep in range(12):
            x.copy_((torch.randn(M, K, device=DEV) * (0.03 * (rep + 1))).half())
            g.replay()
            torch.cuda.synchronize()
            y_two = lin(x)
            torch.cuda.synchronize()
            assert torch.equal(y, y_two), (M, rep)
    finally:
        gemlite_amd.core.FUSE_ACT_QUANT_ROWS = False


def test_warmup_sizes_the_workspace_and_leaves_nothing_to_the_first_request():
    """helper.warmup(processor, shapes, batch_sizes): one call per shape and batch size on a random layer — afterwards the stream's workspace
    holds the largest plan of those shapes (no regrow, hence no allocation, inside a later graph capture)."""
    H = gemlite_amd.helper
    dev = torch.device(DEV)
    stream = _hip.current_stream_handle(dev)
    H.warmup(H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16), shapes=[(2048, 4096)], batch_sizes=[1, 16, 300, 1024])
    ws = _hip.workspace(dev, stream, 0)
    lin = H.A8W8_int8_dynamic(device=DEV, dtype=torch.float16).from_weights((torch.randn(2048, 4096) / 30).half())
    for M in (1, 16, 300, 1024):
        y = lin((torch.randn(M, 4096, device=DEV) / 10).half())
        assert tuple(y.shape) == (M, 2048)
    torch.cuda.synchronize()
    assert _hip.workspace(dev, stream, 0).data_ptr() == ws.data_ptr()  # same buffer: nothing grew


def test_shipped_tuning_table_autoloads_by_device_and_its_entries_stay_correct():
    """gemlite_amd/configs/mi355x.json is found from the device (name or ISA), loaded once on the first launch like the
    reference's configs/<gpu>.json (core.py:634-654), and launches that hit an entry still match the oracle."""
    import ast, json
    from gemlite_amd import core
    path = core.get_default_cache_config(0)
    assert path is not None and path.endswith("mi355x.json")
    table = json.load(open(path))
    saved = {k: dict(v) for k, v in core.GEMLITE_HIP_CONFIG_CACHE.items()}
    try:
        core.GemLiteLinear.reset_config()
        core._AUTOLOAD_DONE = False
        assert core.autoload_default_config(0) == path
        assert all(k in core.GEMLITE_HIP_CONFIG_CACHE.get(f, {}) for f, e in table.items() for k in e)
        done = 0
        for fam, entries in table.items():
            for key, e in list(entries.items())[:3]:
                Mb, N, K, gs, eps, tid = ast.literal_eval(key)
                if N * K > 5120 * 13824:
                    continue  # oracle time
                lin = _make_layer(N, K, 4, gs, torch.float16, seed=70 + done)
                x = torch.from_numpy(O.gen_x(Mb, K, seed=Mb)).to(DEV)
                a = core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
                assert core.lookup_tuning(-1, Mb, a) == tuple(e["tuning"])
                y = lin(x)
                torch.cuda.synchronize()
                cols = slice(0, 256)
                _compare(f"table/{fam}/{key}", y[:, cols], torch.from_numpy(_oracle_columns(lin, x, cols)), 1,
                         extra=dict(tuning=e["tuning"]))
                done += 1
        assert done >= 3
    finally:
        core.GemLiteLinear.reset_config()
        core.GEMLITE_HIP_CONFIG_CACHE.update(saved)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [4, 2])
def test_a8wn_fp8_activations_on_the_mfma_kernel(nbits, tdt):
    """A8Wn dynamic (helper.py:502-615): fp8 e4m3 activations (per-token scales) x packed 4- / 2-bit grouped weights.  The
    dequantised weight is rounded to e4m3 before the dot like the reference's `b.to(a.dtype)` (gemm_kernels.py:384) and the
    product runs on v_mfma_f32_32x32x16_fp8_fp8: every tile height, ragged M, uneven K slices, grouped (3, 2) and
    channel-wise post-scale (1, 3) modes, against the float64 oracle on the same fp8-rounded operands."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    N, K = 1024, 1280   # (N / 16 >= 64 blocks: the decode kernel applies)
    out_code = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
    for gs, post in ((128, False), (K, True), (K, False)):
        W_q, sc, zr = O.gen_data(N, K, nbits, gs, seed=80 + nbits)
        lin = H.A8Wn_HQQ_INT_dynamic(device=DEV, dtype=tdt, post_scale=post, W_nbits=nbits).from_weights(
            torch.from_numpy(W_q), torch.from_numpy(sc).to(tdt), torch.from_numpy(zr).to(tdt))
        assert (lin.W_group_mode, lin.channel_scale_mode) == ((3, 2) if gs == 128 or not post else (1, 3))
        for mi, M in ((1, 1), (1, 2), (1, 3), (1, 16), (1, 29), (2, 50), (2, 64), (4, 100), (8, 300)):
            x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
            xq_t, sx_t = scale_activations_per_token(x, w_dtype=torch.float8_e4m3fn)
            xq, sx = O.scale_activations_per_token(x, O.FP8E4)
            assert np.array_equal(xq_t.float().cpu().numpy().astype(np.float64), xq)  # activation quant is bit-exact
            z = O.to_f64(lin.zeros.data)
            y_or = O.forward_packed(xq, lin.W_q.data.cpu().numpy(), O.to_f64(lin.scales.data), z, W_nbits=nbits,
                                    group_size=lin.group_size, W_group_mode=lin.W_group_mode,
                                    channel_scale_mode=lin.channel_scale_mode, scales_x=sx, weight_cast_code=O.FP8E4)
            if M <= 4:  # decode sizes: the GEMV-class kernel (K not split across blocks).  2 .. 4 rows: per-weight cast to e4m3 like
                # the reference's GEMM_SPLITK family; ONE row = its GEMV family, whose dot product keeps the dequantised weight in the
                # metadata type (gemv_revsplitK_kernels.py:331-332) — round 4, pinned by the reference's own M = 1 output on the
                # MI355X (tests/golden/fullsize_ref_r4.npz, a8w4_fp8dyn_m1)
                # (2 .. 4 rows: the default is the rows kernel since round 4 — tuning[0] = 7 keeps this one)
                y = _hip_matmul(xq_t, lin.W_q, lin.scales, lin.zeros, sx_t, lin.get_meta_args(), -1, (7, 0, 0, 0) if M > 1 else (0, 0, 0, 0))
                torch.cuda.synchronize()
                y_or1 = y_or if M > 1 else O.forward_packed(
                    xq, lin.W_q.data.cpu().numpy(), O.to_f64(lin.scales.data), z, W_nbits=nbits, group_size=lin.group_size,
                    W_group_mode=lin.W_group_mode, channel_scale_mode=lin.channel_scale_mode, scales_x=sx, weight_cast_code=out_code)
                _compare(f"a8wn/w{nbits}/{str(tdt)[6:]}/g{gs}/post{int(post)}/M{M}/gemv", y, y_or1, out_code, abs_gate=None)
            if 2 <= M <= 64:  # round 4: the rows kernel (the default from 2 rows): per-weight cast to e4m3 like the tile kernel
                tun = (0, 0, 0, 0)
                a = gemlite_amd.core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
                a.matmul_type, a.M, a.x, a.out, a.scales_x = -1, M, 0x1000, 0x1000, 0x1000
                a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = K, 1, N, 1
                a.input_dtype = lin.input_dtype.value
                for i in range(4):
                    a.tuning[i] = tun[i]
                name = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
                assert name == f"a8w{nbits}_rows_kernel<%s>" % ("16x16" if M <= 16 else ("32x16" if M <= 32 else "64x16")), (M, name)
                y = _hip_matmul(xq_t, lin.W_q, lin.scales, lin.zeros, sx_t, lin.get_meta_args(), -1, tun)
                torch.cuda.synchronize()
                _compare(f"a8wn/w{nbits}/{str(tdt)[6:]}/g{gs}/post{int(post)}/M{M}/rows", y, y_or, out_code, abs_gate=None)
            for sk in (0, 3):
                tuning = (0, sk, mi, 0)
                y = _hip_matmul(xq_t, lin.W_q, lin.scales, lin.zeros, sx_t, lin.get_meta_args(), -1, tuning)
                torch.cuda.synchronize()
                # which kernel ran: ask with the same tuning
                a = gemlite_amd.core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
                a.matmul_type, a.M, a.x, a.out, a.scales_x = -1, M, 0x1000, 0x1000, 0x1000
                a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = K, 1, N, 1
                a.input_dtype = lin.input_dtype.value
                for i in range(4):
                    a.tuning[i] = tuning[i]
                name = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
                assert name == f"gemm_a8w{nbits}_mma_kernel<{32 * mi}x128>", name
                _compare(f"a8wn/w{nbits}/{str(tdt)[6:]}/g{gs}/post{int(post)}/M{M}/sk{sk}", y, y_or, out_code, abs_gate=None,
                         extra=dict(kernel=name))


@pytest.mark.parametrize("M", [1, 2, 4, 7, 16, 40, 64, 200])
def test_bitnet_int8_activations_on_the_int8_mfma_are_exact(M):
    """A8W158_INT_dynamic (helper.py:1006-1062): int8 activations x ternary 2-bit codes with a scalar zero of 1, fp32
    per-tensor scale, modes (1, 3).  (q - 1) is exact int8, v_mfma_i32_32x32x16_i8 accumulates exact int32: the result must
    equal fp16(int_dot * s_x * s_w) computed in fp32 — bit for bit."""
    H = gemlite_amd.helper
    N, K = 1024, 2560
    torch.manual_seed(M)
    Wt = torch.randint(-1, 2, (N, K)).half()
    lin = H.A8W158_INT_dynamic(device=DEV).from_weights(Wt, torch.tensor(0.02))
    x = (torch.randn(M, K) / 10).half().to(DEV)
    y = lin(x)
    torch.cuda.synchronize()
    # (round 4: 2 .. 64 rows on a8wn_rows_kernel — v_mfma_i32_16x16x64_i8 — the 8-wave tile kernel above)
    assert _kernel_name(lin, x).startswith("gemv_a8w2_kernel<" if M <= 1 else ("a8w2_rows_kernel<" if M <= 64 else "gemm_a8w2_mma_kernel<")), _kernel_name(lin, x)
    xq, sx = scale_activations_per_token(x, w_dtype=torch.int8)
    dot = (xq.cpu().to(torch.int64) @ Wt.to(torch.int64).t())  # exact
    assert int(dot.abs().max()) < (1 << 24)
    sw = lin.scales.data.float().cpu().reshape(1, -1)
    ref = (dot.float() * (sx.cpu().float() * sw)).half()  # the epilogue's operation order (epilogue_scale)
    assert torch.equal(y.cpu(), ref)


def test_coverage_kernel_is_correct_and_announced_once(caplog):
    """A shape no specialised kernel takes (N % 64 != 0) still runs — on the coverage kernel — and the host says so once
    per shape ('orders of magnitude slower with no warning')."""
    import logging
    from gemlite_amd import core
    N, K = 200, 512
    lin = _make_layer(N, K, 4, 128, torch.float16, seed=90)
    x = torch.from_numpy(O.gen_x(3, K, seed=3)).to(DEV)
    assert _kernel_name(lin, x) == "generic_matmul_kernel"
    core._COVERAGE_CHECKED.clear()
    with caplog.at_level(logging.WARNING, logger=core.logger.name):
        y = lin(x)
        y2 = lin(x)
    torch.cuda.synchronize()
    hits = [r for r in caplog.records if "coverage kernel" in r.getMessage()]
    assert len(hits) == 1 and "N=200" in hits[0].getMessage()
    assert torch.equal(y, y2)
    _compare("coverage/200x512/M3", y, _oracle_from_layer(lin, x), 1)


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        nbits = int(rng.choice([4, 4, 4, 2, 2, 8, 1]))
        K = int(rng.choice([256, 384, 512, 640, 896, 1024, 1280, 1536, 2048, 2816]))
        N = int(rng.choice([64, 128, 192, 256, 384, 512, 640, 1024, 1408]))
        gs = int(rng.choice([64, 128, 128, 128, 256, K]))
        if K % gs:
            gs = 128
        M = int(rng.choice([1, 1, 2, 3, 5, 8, 13, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256, 300, 513]))
        tdt = torch.float16 if rng.random() < 0.5 else torch.bfloat16
        zk = str(rng.choice(["tensor", "tensor", "tensor", "none", "int"]))
        fma = bool(rng.random() < 0.7)
        out.append((i, nbits, N, K, gs, M, tdt, zk, fma))
    return out


@pytest.mark.parametrize("case", _random_cases(72, seed=2024), ids=lambda c: f"r{c[0]}-w{c[1]}-{c[2]}x{c[3]}-g{c[4]}-M{c[5]}-{str(c[6])[6:]}-{c[7]}-{'fma' if c[8] else 'sub'}")
def test_random_shapes_dtypes_and_modes_against_the_oracle(case):
    """Seeded random sweep over bit width x (N, K) (including K = 128 * odd and N = 64 * odd) x group size x M (all kernel
    families and their thresholds) x dtype x zero kind x dequant mode: whatever kernel the planners pick — their choices
    come from a fitted cost model, so odd shapes reach (tile, K-slice) pairs no hand-written test names — the result
    matches the float64 oracle on the tensors the layer holds."""
    i, nbits, N, K, gs, M, tdt, zk, fma = case
    try:
        lin = _make_layer(N, K, nbits, gs, tdt, seed=1000 + i, zeros_kind=zk, fma=fma)
    except (NotImplementedError, ValueError) as e:  # a combination the reference's constructor / pack() rejects as well
        pytest.skip(f"rejected at construction: {e}")
    x = torch.from_numpy(O.gen_x(M, K, seed=i).astype(np.float32)).to(tdt).to(DEV)
    name = _kernel_name(lin, x)
    y = lin(x)
    torch.cuda.synchronize()
    assert y.shape == (M, N)
    # 8-bit codes and shift-only modes give |y| >> 1: the absolute gate of the 4-bit fixtures does not apply
    _compare(f"random/{i}", y, _oracle_from_layer(lin, x), lin.output_dtype.value, abs_gate=None,
             extra=dict(kernel=name, case=[nbits, N, K, gs, M, str(tdt), zk, fma]))


def test_cpp_eager_host_path_matches_the_python_path_and_notices_every_change():
    """gemlite_amd/_fast.so (round 4): `layer(x)` without Python between tensor and launch.  It must return the Python path's bits, and
    fall back (never use stale pointers / stale tuning) when the layer's tensors are swapped or re-allocated in place, when the tuning
    table changes, for non-contiguous or wrong-dtype x, and keep bias / batched shapes."""
    from gemlite_amd import core
    if core._FAST is None:
        pytest.skip("gemlite_amd/_fast.so not built")
    lin = _make_layer(512, 1024, 4, 128, torch.float16, seed=91)
    x = torch.from_numpy(O.gen_x(3, 1024, seed=2)).to(DEV)

    def slow(layer, xx):
        return core._forward_impl(xx, layer.bias, layer.get_tensor_args(), layer.get_meta_args(), -1)
    y0 = lin(x)                      # slow call, installs the handle
    assert lin.__dict__.get("_fast") is not None
    y1 = lin(x)                      # fast call
    assert torch.equal(y0, y1) and torch.equal(y1, slow(lin, x))
    # batched input + bias
    lin.bias = torch.randn(512, dtype=torch.float16, device=DEV)
    assert lin.__dict__.get("_fast") is None          # assigning a template field drops the handle
    xb = torch.from_numpy(O.gen_x(6, 1024, seed=3)).to(DEV).view(2, 3, 1024)
    ya = lin(xb)
    yb = lin(xb)
    assert lin.__dict__.get("_fast") is not None and yb.shape == (2, 3, 512) and torch.equal(ya, yb) and torch.equal(yb, slow(lin, xb))
    # tensors re-allocated IN PLACE (what module.to() / param.data = ... do): same Python objects, new storage
    lin.W_q.data = lin.W_q.data.clone()
    lin.scales.data = (lin.scales.data.float() * 2).to(lin.scales.dtype)
    y2 = lin(x)
    assert torch.equal(y2, slow(lin, x)) and not torch.equal(y2[:, :8], (y1 + lin.bias)[:, :8])
    y2b = lin(x)
    assert torch.equal(y2, y2b)
    # the tuning table changes: handles of the old epoch must not be used
    a = core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
    try:
        core.GEMLITE_HIP_CONFIG_CACHE.setdefault("GEMM_SPLITK", {})[core.config_key(3, a.N, a.K, 128, 8, a.type_id)] = {"tuning": [0, 2, 0, 0]}
        y3 = lin(x)
        y3b = lin(x)     # fast again, with the looked-up tuning of this M
        assert torch.equal(y3, y3b) and torch.equal(y3, slow(lin, x))
    finally:
        core.GemLiteLinear.reset_config()
    # x the fast path does not take: non-contiguous, other dtype (-> the Python path's behaviour, including its errors)
    xnc = torch.from_numpy(O.gen_x(3, 2048, seed=4)).to(DEV)[:, ::2]
    assert torch.equal(lin(xnc), slow(lin, xnc.contiguous()))
    with pytest.raises(Exception):
        lin(torch.zeros(3, 1000, dtype=torch.float16, device=DEV))   # wrong K: the Python path's ValueError
    torch.cuda.synchronize()


@pytest.mark.parametrize("amp", [1e-3, 3e-4, 1e-4])
def test_small_magnitude_fp16_activations_on_the_decode_kernels(amp):
    """the fp16 decode kernels of rounds 2-3 pre-scale half of the x pairs by 2^-4 (4-bit; 2^-6 for 2-bit fields) IN fp16 so
    that one AND yields two products; for |x| below 2^-10 that scaled value is an fp16 subnormal and loses mantissa bits (the reference
    multiplies the unscaled x — but accumulates its GEMV in fp16, where products of this size are subnormal as well).  The round-4 decode
    kernel (gemv_decode.hip) keeps the two field positions in separate fp32 accumulators instead and applies the 2^-4 once to the sum:
    nothing is rounded there, so its error against the float64 oracle is the fp16 rounding of the OUTPUT alone whatever the magnitude
    of x (round 5: the round-3 decode kernel and the 4-bit forms of gemv_wn_kernel do the same).  Round 6: ONE bound for
    every decode kernel — the matrix-core decode kernels build their fragments so that every code of one MFMA sits at the same bit offset
    and keep one fp32 accumulator per offset (gemv_mfma.hip, `Planes`), the 2- / 1-bit forms of gemv_wn_kernel keep one accumulator set per
    field position: nothing pre-scales x in fp16 any more."""
    from gemlite_amd.core import _hip_matmul
    for nbits, N, K, M, tuning, want, bound in (
            (4, 1024, 4096, 1, (0, 0, 0, 0), "gemv_w4_decode3_kernel", 4e-4),          # exact products: output rounding only
            (4, 4096, 4096, 1, (0, 0, 0, 4096), "gemv_w4_decode_kernel", 4e-4),       # round 5: split accumulators there too
            (4, 4096, 4096, 1, (0, 0, 0, 16), "gemv_wn_kernel<tile16,xdirect,16w>", 4e-4),   # ... and in the 4-bit forms of gemv_wn_kernel: x direct,
            (4, 4096, 8192, 1, (0, 0, 0, 512), "gemv_wn_kernel<tile16>", 4e-4),              #     x staged through LDS
            (4, 8192, 4096, 1, (0, 0, 0, 1024), "gemv_mfma_kernel", 4e-4),                   # round 6: exact planes on the matrix core (forced: one row of
            (4, 8192, 8192, 1, (0, 0, 0, 1024), "gemv_mfma_kernel", 4e-4),                   #     4-bit words defaults to the dot-product family since round 6)
            (4, 8192, 8192, 1, (0, 0, 0, 0), "gemv_wn_kernel<tile32>", 4e-4),                #     ... the default of the 8192^2 decode shape of the bench line
            (4, 1024, 4096, 3, (0, 0, 0, 0), "gemv_mfma_kernel", 4e-4),                      #     2 .. 4 rows
            (4, 16384, 4096, 1, (24, 0, 0, 1024), "gemv_mfma_kernel<tile64>", 4e-4),         #     64-column tiles
            (2, 1024, 4096, 1, (0, 0, 0, 1024), "gemv_w2_mfma_kernel", 4e-4),                #     2-bit words: four planes
            (2, 4096, 8192, 1, (24, 0, 0, 1024), "gemv_w2_mfma_kernel<tile64>", 4e-4),
            (2, 4096, 8192, 1, (0, 0, 0, 512), "gemv_wn_kernel", 4e-4),                      # round 6: four accumulator sets (BASELINE configs[4]'s decode kernel)
            (1, 4096, 4096, 1, (0, 0, 0, 0), "gemv_wn_kernel", 4e-4)):                       # ... eight (1-bit words)
        lin = _make_layer(N, K, nbits, 128, torch.float16, seed=95 + nbits)
        g = np.random.default_rng(7)
        x = torch.from_numpy((g.standard_normal((M, K)) * amp).astype(np.float16)).to(DEV)
        name = _kernel_name(lin, x, -1, tuning)
        assert name.startswith(want), name
        y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tuning)
        torch.cuda.synchronize()
        y_or = _oracle_from_layer(lin, x)
        rel = float(np.abs(y.float().cpu().numpy().astype(np.float64) - y_or).mean() / np.abs(y_or).mean())
        print(f"small-x amp={amp} {name}: rel={rel:.3e}")
        # (the bound is the rounding of the OUTPUT: where mean |y| comes near fp16's subnormal range — 1-bit weights under |x| ~ 1e-4 — the output's
        #  absolute quantum 2^-24 joins the relative half-ulp)
        assert rel < bound + 2.0 ** -25 / np.abs(y_or).mean(), (amp, name, rel, bound)


# ------------------------------------------------------------------------------------------------ round 5: the decode-shaped rows kernel
ROWS5 = (9, 0, 0, 0)  # tuning[0] = 9 forces gemm_w4_rows_kernel


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("gs", [128, 64, 32, 512, 2048])
def test_rows5_kernel_row_tiles_and_group_sizes(gs, tdt):
    """gemm_w4_rows_kernel (gemm_wn_rows.hip): every row-tile count (16 .. 64 rows), ragged M, 64-row blocks along grid.y, groups of
    32 / 64 / 128 / 512 / K (one metadata row), K of 8 chunks (one per wave) — against the oracle."""
    from gemlite_amd.core import _hip_matmul
    N, K = 1024, 2048
    lin = _make_layer(N, K, 4, gs, tdt, seed=50 + gs % 7, scales_kind="group" if gs < K else "channel")
    for M in (2, 5, 16, 17, 31, 33, 48, 49, 64, 100):
        x = torch.from_numpy(O.gen_x(M, K, seed=M + 11).astype(np.float32)).to(tdt).to(DEV)
        y_or = _oracle_from_layer(lin, x)
        for tuning in (ROWS5,):
            name = _kernel_name(lin, x, 3, tuning)
            assert name.startswith("gemm_w4_rows_kernel<"), (name, M, gs, tuning)
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 3, tuning)
            torch.cuda.synchronize()
            _compare(f"rows5/g{gs}/{str(tdt)[6:]}/M{M}/{tuning}", y, y_or, lin.output_dtype.value, extra=dict(kernel=name))


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("gs", [128, 64, 32, 1024, 2048])
def test_rows5_kernel_two_bit_words(gs, tdt):
    """The rows kernel on 2-bit words (A16W2, BitNet A16W158; late round 5): a chunk is 512 k = sixteen k-steps, each lane spreads its half of a
    word to eight nibbles.  Every row-tile count the group size allows (64 / 32 / 16 rows per block, more along grid.y), ragged M, K of four
    chunks (half of the waves idle) and of 10 (two waves with a second chunk) — against the oracle; and what the planner does by default:
    groups of 32 and N % 64 != 0 at M >= 2 come here instead of the coverage kernel."""
    from gemlite_amd.core import _hip_matmul
    for (N, K) in ((1024, 2048), (528, 5120)):
        if K % gs:
            continue
        lin = _make_layer(N, K, 2, gs, tdt, seed=60 + gs % 7, scales_kind="group" if gs < K else "channel")
        for M in (2, 5, 16, 17, 31, 33, 48, 64, 100):
            x = torch.from_numpy(O.gen_x(M, K, seed=M + 13).astype(np.float32)).to(tdt).to(DEV)
            name = _kernel_name(lin, x, 3, ROWS5)
            assert name.startswith("gemm_w2_rows_kernel<"), (name, M, gs)
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 3, ROWS5)
            torch.cuda.synchronize()
            _compare(f"rows5-w2/{N}x{K}g{gs}/{str(tdt)[6:]}/M{M}", y, _oracle_from_layer(lin, x), lin.output_dtype.value, extra=dict(kernel=name))
            if gs == 32 or N % 64 != 0:
                # (round 6: groups of 32 above 32 rows over N % 128 == 0 run on the tile kernel's g32 form)
                g32_tiles = gs == 32 and M > 32 and N % 128 == 0 and K % 256 == 0
                want = "gemm_w2_mma_kernel<32x128,g32>" if g32_tiles else "gemm_w2_rows_kernel<"
                assert _kernel_name(lin, x).startswith(want), (M, _kernel_name(lin, x))


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_rows5_kernel_bitnet_fp32_post_scale(tdt):
    """BitNet A16W158 (ternary weights as 2-bit codes, scalar zero 1, ONE fp32 scale applied per output channel after the K reduction —
    modes (1, 1) with fp32 metadata): the rows kernel's epilogue reads fp32 channel scales, so 2 .. 64 rows of a 4096-wide BitNet layer
    run on it by default."""
    H = gemlite_amd.helper
    g = torch.Generator().manual_seed(77)
    N, K = 4096, 2048
    Wt = torch.randint(-1, 2, (N, K), generator=g).to(tdt)
    lin = H.A16W158_INT(device=DEV, dtype=tdt).from_weights(Wt, torch.tensor(0.02))
    for M in (8, 16, 40, 64):
        x = (torch.randn(M, K, generator=g) / 10).to(tdt).to(DEV)
        name = _kernel_name(lin, x)
        assert name.startswith("gemm_w2_rows_kernel<"), (M, name)
        y = lin(x)
        torch.cuda.synchronize()
        _compare(f"rows5-bitnet/{str(tdt)[6:]}/M{M}", y, _oracle_from_layer(lin, x), lin.output_dtype.value, extra=dict(kernel=name))


@pytest.mark.parametrize("zeros_kind,fma,scales_kind", [("tensor", False, "group"), ("none", True, "group"), ("int", True, "group"), ("tensor", True, "channel")])
def test_rows5_kernel_two_bit_words_all_modes(zeros_kind, fma, scales_kind):
    from gemlite_amd.core import _hip_matmul
    lin = _make_layer(2048, 4096, 2, 128 if scales_kind == "group" else 4096, torch.float16, seed=6, zeros_kind=zeros_kind, fma=fma, scales_kind=scales_kind)
    for M in (6, 24, 56):
        x = torch.from_numpy(O.gen_x(M, 4096, seed=M).astype(np.float32)).half().to(DEV)
        assert _kernel_name(lin, x, -1, ROWS5).startswith("gemm_w2_rows_kernel<"), _kernel_name(lin, x, -1, ROWS5)
        y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, ROWS5)
        torch.cuda.synchronize()
        _compare(f"rows5-w2-modes/{zeros_kind}-{fma}-{scales_kind}/M{M}", y, _oracle_from_layer(lin, x), lin.output_dtype.value)


@pytest.mark.parametrize("zeros_kind,fma,scales_kind", [("tensor", False, "group"), ("none", True, "group"),
                                                         ("int", True, "group"), ("int", True, "channel"),
                                                         ("tensor", True, "channel"), ("none", True, "channel")])
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_rows5_kernel_all_modes(zeros_kind, fma, scales_kind, tdt):
    from gemlite_amd.core import _hip_matmul
    lin = _make_layer(2048, 4096, 4, 128 if scales_kind == "group" else 4096, tdt, seed=5, zeros_kind=zeros_kind, fma=fma,
                      scales_kind=scales_kind)
    for M in (6, 24, 56):
        x = torch.from_numpy(O.gen_x(M, 4096, seed=M).astype(np.float32)).to(tdt).to(DEV)
        assert _kernel_name(lin, x, -1, ROWS5).startswith("gemm_w4_rows_kernel<"), _kernel_name(lin, x, -1, ROWS5)
        y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, ROWS5)
        torch.cuda.synchronize()
        _compare(f"rows5-modes/{zeros_kind}-{fma}-{scales_kind}/{str(tdt)[6:]}/M{M}", y, _oracle_from_layer(lin, x), lin.output_dtype.value)
        # two 16-column tiles per block (tuning[1] = 2; automatic for 4096 < N <= 8192): the same arithmetic per tile, bit for bit
        assert _kernel_name(lin, x, -1, (9, 2, 0, 0)).endswith("x32>") and _kernel_name(lin, x, -1, (9, 1, 0, 0)).endswith("x16>")
        y2 = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (9, 2, 0, 0))
        y1 = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (9, 1, 0, 0))
        torch.cuda.synchronize()
        assert torch.equal(y1, y2) and torch.equal(y, y1), (M, float((y1.float() - y2.float()).abs().max()))


@pytest.mark.parametrize("N,K,gs", [(4096, 4096, 128), (1040, 2816, 128), (512, 256, 128), (2064, 11008, 64), (1024, 8192, 32),
                                    (8192, 4096, 128), (6160, 2048, 64), (11008, 4096, 128), (4352, 11008, 128)])
def test_rows5_kernel_shapes_strides_and_defaults(N, K, gs):
    """Shapes: the headline layer, N % 64 != 0 with K = 11 chunks, a single chunk (7 of 8 waves idle), K = 11008 (43 chunks: 5 or 6 per wave),
    four chunks per wave; more tiles than CUs (512 / 385 / 688 / 272 blocks); a strided x (rows of a wider buffer); the kernel agrees with the round-4
    kernels within both tolerances; and what the planner does by default: the rows kernel inside its budget, for groups of 32 and for
    N % 64 != 0 at any M >= 2."""
    from gemlite_amd.core import _hip_matmul
    tdt = torch.float16
    lin = _make_layer(N, K, 4, gs, tdt, seed=N % 13)
    for M in (8, 32, 64):
        wide = torch.from_numpy(O.gen_x(M, K + 64, seed=M + 5).astype(np.float32)).to(tdt).to(DEV)
        for x in (wide[:, :K].contiguous(), wide[:, 32:32 + K]):  # contiguous / row stride K + 64, 64-byte offset
            name = _kernel_name(lin, x, -1, ROWS5)
            assert name.startswith("gemm_w4_rows_kernel<"), name
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, ROWS5)
            torch.cuda.synchronize()
            _compare(f"rows5-shapes/{N}x{K}g{gs}/M{M}/{'c' if x.is_contiguous() else 's'}", y, _oracle_from_layer(lin, x), 1, extra=dict(kernel=name))
        y4 = _hip_matmul(x.contiguous(), lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, (0, 0, 0, 65536))  # the round-4 choice
        torch.cuda.synchronize()
        _compare(f"rows5-vs-r4/{N}x{K}g{gs}/M{M}", y, y4.float().cpu().numpy(), 1)
    x = torch.from_numpy(O.gen_x(16, K, seed=1).astype(np.float32)).to(tdt).to(DEV)
    if gs == 32 or N % 64 != 0:
        for M in (2, 16, 200):
            xm = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
            g32_tiles = gs == 32 and M > 32 and N % 128 == 0 and K % 256 == 0   # round 6: the tile kernel's g32 form above 32 rows
            assert _kernel_name(lin, xm).startswith("gemm_w4_mma_kernel<32x128,g32>" if g32_tiles else "gemm_w4_rows_kernel<"), (M, _kernel_name(lin, xm))
            y = lin(xm)
            torch.cuda.synchronize()
            _compare(f"rows5-only-here/{N}x{K}g{gs}/M{M}", y, _oracle_from_layer(lin, xm), 1)
    assert not _kernel_name(lin, x, -1, (0, 0, 0, 65536)).startswith("gemm_w4_rows_kernel")


def test_rows5_kernel_is_deterministic_graph_capturable_and_linear():
    from gemlite_amd.core import _hip_matmul
    lin = _make_layer(4096, 4096, 4, 128, torch.float16, seed=3)
    x = torch.from_numpy(O.gen_x(32, 4096, seed=2).astype(np.float32)).half().to(DEV)
    f = lambda v: _hip_matmul(v, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, ROWS5)
    y0 = f(x)
    for _ in range(3):
        assert torch.equal(f(x), y0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f(x)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            yg = f(x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y0)
    assert torch.count_nonzero(f(torch.zeros_like(x))) == 0
    y2 = f(x * 2)
    assert torch.allclose(y2.float(), 2 * y0.float(), rtol=2e-3, atol=2e-3)


# ------------------------------------------------------------------------------------------------ round 6: groups of 32 on the tile kernel
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [1, 2, 4, 8])
def test_mma_kernel_groups_of_32(nbits, tdt):
    """Round 6: groups of 32 on the 8-wave MFMA kernel — two (scale, zero) pairs per column and 64-k sub-block, 32-row
    tiles (template parameter NGS = 2).  1- / 8-bit packed words at M >= 2 ran on the coverage kernel until round 5 (and 2-bit words whose K is
    not a multiple of 512); 4- / 2-bit words are forced onto it here (matmul_type GEMM + tuning[3] & 65536 skips the rows kernel).  Ragged M,
    several row and column tiles, K of 5 steps (uneven slices), forced split-K, every W_group_mode — against the float64 oracle."""
    from gemlite_amd.core import _hip_matmul
    N, K = 256, 1280
    for zeros_kind, fma in (("tensor", True), ("tensor", False), ("none", True)):
        lin = _make_layer(N, K, nbits, 32, tdt, seed=50 + nbits, zeros_kind=zeros_kind, fma=fma)
        for M in (2, 29, 64, 100, 300):
            x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
            y_or = _oracle_from_layer(lin, x)
            for sk in (0, 1, 3):
                tuning = (0, sk, 0, 65536)
                name = _kernel_name(lin, x, 4, tuning)
                assert name == f"gemm_w{nbits}_mma_kernel<32x128,g32>", name
                y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 4, tuning)
                torch.cuda.synchronize()
                _compare(f"mma_g32/w{nbits}/{str(tdt)[6:]}/{zeros_kind}{int(fma)}/M{M}/sk{sk}", y, y_or, lin.output_dtype.value,
                         abs_gate=None if nbits == 8 else 1e-3, extra=dict(kernel=name))
    # what the planner does by default with these layers at M >= 2: no coverage kernel
    lin = _make_layer(512, 1024, nbits, 32, tdt, seed=3)
    for M in (2, 16, 64, 200):
        x = torch.from_numpy(O.gen_x(M, 1024, seed=M).astype(np.float32)).to(tdt).to(DEV)
        name = _kernel_name(lin, x)
        assert "generic" not in name, (M, name)
        y = lin(x)
        torch.cuda.synchronize()
        _compare(f"mma_g32/default/w{nbits}/{str(tdt)[6:]}/M{M}", y, _oracle_from_layer(lin, x), lin.output_dtype.value,
                 abs_gate=None if nbits == 8 else 1e-3, extra=dict(kernel=name))


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nbits", [4, 2, 1, 8])
def test_group_sizes_that_are_not_a_power_of_two(nbits, tdt):
    """Round 6: the reference admits any group size that divides K (core.py:253-271; per-block scale loads
    gemm_splitK_kernels.py:391-405).  Multiples of 32 that are not a power of two ran on the coverage kernel until round 5; now the 8-wave tile kernel
    takes them at every M — the metadata row of a slice is k / group by one scalar multiply-high.  Groups of 96 / 160 / 192 / 384 / 768 on the form with two
    metadata pairs per 64-k sub-block (32-row tiles: the only instantiations that carry the division — the power-of-two forms lost 0.6 us at cfgA with it), ragged M,
    forced split-K, every W_group_mode; against the float64 oracle."""
    from gemlite_amd.core import _hip_matmul
    for gs, N, K in ((96, 256, 1536), (160, 128, 1280), (192, 256, 1536), (384, 256, 3072), (768, 512, 3072)):
        for zeros_kind, fma in (("tensor", True), ("tensor", False), ("none", True)):
            lin = _make_layer(N, K, nbits, gs, tdt, seed=60 + nbits + gs, zeros_kind=zeros_kind, fma=fma)
            for M in (1, 2, 29, 100, 300):
                x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
                y_or = _oracle_from_layer(lin, x)
                for sk in ((0, 3) if M in (29, 300) else (0,)):
                    tuning = (0, sk, 0, 0)
                    name = _kernel_name(lin, x, -1, tuning)
                    if M == 1 and name.startswith("gemv_"):   # one row: the dot-product GEMV where its chunk geometry fits (a lane's k span inside one group)
                        assert sk == 0 and "decode3" not in name, name
                    else:
                        assert name.startswith(f"gemm_w{nbits}_mma_kernel<"), (gs, M, name)
                        assert "g32" in name, (gs, name)   # (the NGS = 2 form is the one that divides: every group size that is not a power of two runs there)
                    y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), -1, tuning)
                    torch.cuda.synchronize()
                    _compare(f"npot/w{nbits}/{str(tdt)[6:]}/g{gs}/{zeros_kind}{int(fma)}/M{M}/sk{sk}", y, y_or, lin.output_dtype.value,
                             abs_gate=None if nbits == 8 else 1e-3, extra=dict(kernel=name))


@pytest.mark.parametrize("nbits", [4, 2])
def test_a8wn_fp8_activations_groups_of_32(nbits):
    """A8Wn dynamic with groups of 32 (coverage kernel until round 5 above the rows kernel's 64 rows): fp8 e4m3 activations x packed words with
    two metadata pairs per sub-block, against the float64 oracle on the same fp8-rounded operands."""
    from gemlite_amd.core import _hip_matmul
    H = gemlite_amd.helper
    tdt = torch.float16
    N, K = 512, 1280
    out_code = gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt].value
    W_q, sc, zr = O.gen_data(N, K, nbits, 32, seed=90 + nbits)
    lin = H.A8Wn_HQQ_INT_dynamic(device=DEV, dtype=tdt, post_scale=False, W_nbits=nbits).from_weights(
        torch.from_numpy(W_q), torch.from_numpy(sc).to(tdt), torch.from_numpy(zr).to(tdt))
    for M in (70, 100, 300):
        x = torch.from_numpy(O.gen_x(M, K, seed=M).astype(np.float32)).to(tdt).to(DEV)
        xq_t, sx_t = scale_activations_per_token(x, w_dtype=torch.float8_e4m3fn)
        xq, sx = O.scale_activations_per_token(x, O.FP8E4)
        y_or = O.forward_packed(xq, lin.W_q.data.cpu().numpy(), O.to_f64(lin.scales.data), O.to_f64(lin.zeros.data), W_nbits=nbits,
                                group_size=lin.group_size, W_group_mode=lin.W_group_mode,
                                channel_scale_mode=lin.channel_scale_mode, scales_x=sx, weight_cast_code=O.FP8E4)
        for sk in (0, 2):
            tuning = (0, sk, 0, 0)
            a = gemlite_amd.core._static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
            a.matmul_type, a.M, a.x, a.out, a.scales_x = -1, M, 0x1000, 0x1000, 0x1000
            a.stride_xm, a.stride_xk, a.stride_om, a.stride_on = K, 1, N, 1
            a.input_dtype = lin.input_dtype.value
            for i in range(4):
                a.tuning[i] = tuning[i]
            name = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
            assert name == f"gemm_a8w{nbits}_mma_kernel<32x128,g32>", name
            y = _hip_matmul(xq_t, lin.W_q, lin.scales, lin.zeros, sx_t, lin.get_meta_args(), -1, tuning)
            torch.cuda.synchronize()
            _compare(f"a8wn_g32/w{nbits}/M{M}/sk{sk}", y, y_or, out_code, abs_gate=None, extra=dict(kernel=name))
