"""HIP library vs THE REFERENCE'S OWN OUTPUTS at BASELINE sizes.

tests/golden/fullsize_ref.npz holds what /root/reference's Triton kernels produced ON AN MI355X (oracle/make_ref.sh stages the
package under oracle/_ref/, oracle/run_ref_gpu.py --which ref runs it there; committed with the round-3 timings in
profiles/r03/reference_triton_mi355x.json): every 16th column (5, 21, 37, ...) of every output row of cfgA M = 1 / 16 / 256,
cfgB M = 256 bf16, A8W8 int8 / fp8 (OCP e4m3), A16W2 16384^2, FP8 x FP8 16384^2 and the block-scaled processors.  Inputs are
regenerated from the same seeds by the same builders (oracle/run_ref_gpu.py CASES).  Bounds: integer paths bit-exact; paths where
both sides accumulate in fp32 within 1e-4 of the reference's mean |y| (measured 4e-7 .. 1e-5: different summation order only);
the reference's GEMV family accumulates in fp16 (GEMLITE_ACC_DTYPE, core.py:37-52) and carries ~2e-3 of its own rounding.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "fullsize_ref.npz")

# case -> (bound on mean|hip - ref| / mean|ref|, exact)
BOUNDS = {
    "cfgA_fp16_m1": (5e-3, False), "cfgA_fp16_m16": (1e-3, False), "cfgA_fp16_m256": (1e-4, False), "cfgA_bf16_m256": (1e-4, False),
    "cfgB_bf16_m256": (1e-4, False), "cfgB_fp16_m1": (6e-3, False),
    "a8w8_int8_m1": (0.0, True), "a8w8_int8_m16": (0.0, True), "a8w8_int8_m256": (0.0, True),
    "a8w8_fp8_m1": (1e-6, False), "a8w8_fp8_m16": (1e-6, False), "a8w8_fp8_m256": (1e-6, False),
    "a16w2_16384_m1": (8e-3, False), "a16w2_16384_m256": (1e-4, False), "fp8_16384_m256": (1e-6, False),
    "mx_a8w8_m16": (1e-3, False), "mx_a8w8_m256": (4e-3, False), "mx_a8w4_m16": (1e-3, False),
    "mx_a4w4_m16": (1e-5, False), "mx_a4w4_m256": (1e-5, False), "mx_a16w4_m16": (1e-5, False),
}


def _cases():
    from oracle.run_ref_gpu import CASES
    return {name: build for name, build, _ in CASES}


@pytest.mark.parametrize("name", sorted(BOUNDS))
def test_hip_matches_reference_outputs_from_the_mi355x(name):
    import gemlite_amd
    from oracle.run_ref_gpu import COL0, COLSTEP
    gold = np.load(GOLD)
    assert name in gold.files, f"{name} missing from the fixture"
    assert int(gold["col0"]) == COL0 and int(gold["colstep"]) == COLSTEP
    layer, x = _cases()[name](gemlite_amd)
    y = layer(x)
    torch.cuda.synchronize()
    dt = str(gold[name + "__dtype"])
    ref = torch.from_numpy(gold[name]).view({"torch.float16": torch.float16, "torch.bfloat16": torch.bfloat16}[dt]).float().numpy().astype(np.float64)
    got = y[:, COL0::COLSTEP].float().cpu().numpy().astype(np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all()
    bound, exact = BOUNDS[name]
    scale = np.abs(ref).mean()
    rel = np.abs(got - ref).mean() / scale
    if exact:
        assert np.array_equal(got, ref), (name, rel)
    else:
        assert rel < bound, (name, rel, bound)
        # a single output may differ by one rounding step of the output type where the two summation orders straddle a rounding
        # boundary: 2^-8 (bf16) / 2^-11 (fp16) of the LARGEST output, plus the reference's own accumulation noise for that path
        ulp = 2.0 ** -7 if dt == "torch.bfloat16" else 2.0 ** -10
        assert np.abs(got - ref).max() <= ulp * np.abs(ref).max() + 80 * max(bound, 1e-4) * scale, name


def test_fixture_is_the_reference_run_recorded_in_profiles():
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "r03", "reference_triton_mi355x.json")))
    done = {r["case"] for r in rec["report"] if "us" in r}
    assert set(BOUNDS) <= done


# ---- round 4: the kernels added after the first fixture, pinned to the reference's outputs as well ------------------
# tests/golden/fullsize_ref_r4.npz = `python oracle/run_ref_gpu.py --which ref --only r4 --fixture fullsize_ref_r4.npz` on an MI355X
# (timings: profiles/r04/reference_triton_mi355x_r4.json).  Bounds: the reference's GEMV / GEMM_SPLITK families accumulate (fp16
# inputs) or round partial sums (bf16 outputs through atomics) in 16 bits — measured distance of the fp32-accumulating HIP kernels
# 2e-4 (fp16) / 1.7e-3 (bf16) at 2 .. 8 rows, 4.4e-3 at bf16 M = 1; 64 rows was the MFMA tile kernel on both sides in round 4 (3e-7).
GOLD_R4 = os.path.join(ROOT, "tests", "golden", "fullsize_ref_r4.npz")
BOUNDS_R4 = {
    "cfgA_bf16_m1": 9e-3, "a16w2_16384_bf16_m1": 9e-3,
    **{f"cfgA_fp16_m{m}": 1e-3 for m in (2, 4, 8)}, **{f"cfgA_bf16_m{m}": 4e-3 for m in (2, 4, 8)},
    # round 5: 64 rows run on gemm_w4_rows_kernel, which — like the 2 .. 32-row kernels — multiplies RAW integer codes and applies scale / zero
    # to the fp32 group sums instead of rounding every dequantised weight to 16 bits as the reference's GEMM family does: ~2^-12 (fp16) /
    # ~2^-9 (bf16) of mean |y| away from the reference, and closer to the float64 oracle (tests/test_gpu_parity.py)
    "cfgA_fp16_m64": 1e-3, "cfgA_bf16_m64": 4e-3,
    "a8w4_fp8dyn_m1": 9e-3, "a8w4_fp8dyn_m16": 9e-3, "a8w4_fp8dyn_m256": 9e-3,
}


@pytest.mark.parametrize("name", sorted(BOUNDS_R4))
def test_hip_matches_reference_outputs_from_the_mi355x_round4_cases(name):
    import gemlite_amd
    from oracle.run_ref_gpu import COL0, COLSTEP
    gold = np.load(GOLD_R4)
    assert name in gold.files, f"{name} missing from the fixture"
    layer, x = _cases()[name](gemlite_amd)
    y = layer(x)
    torch.cuda.synchronize()
    dt = str(gold[name + "__dtype"])
    ref = torch.from_numpy(gold[name]).view({"torch.float16": torch.float16, "torch.bfloat16": torch.bfloat16}[dt]).float().numpy().astype(np.float64)
    got = y[:, COL0::COLSTEP].float().cpu().numpy().astype(np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all()
    rel = np.abs(got - ref).mean() / np.abs(ref).mean()
    assert rel < BOUNDS_R4[name], (name, rel, BOUNDS_R4[name])


# ---- round 6: the territory of gemm_w4_rows_kernel (round 5), pinned to the reference's outputs ---------------------------
# tests/golden/fullsize_ref_r5.npz = `python oracle/run_ref_gpu.py --which ref --only r5 --fixture fullsize_ref_r5.npz` on an MI355X
# (scripts/r6/run_a.sh; timings + the distances measured on the spot: profiles/r06/reference_triton_mi355x_r5.json, hip_same_method_r5.json).
# Why these bounds are not the GEMM family's 1e-4 (and why `cfgA_*_m64` above moved from 1e-4 when that shape moved to the rows kernel in
# round 5): the reference's GEMM / GEMM_SPLITK kernels round EVERY dequantised weight to the 16-bit type before tl.dot
# (triton_kernels/utils.py:73-87), a relative error of up to 2^-11 (fp16) / 2^-8 (bf16) per weight; the rows kernel multiplies the raw
# integer codes in the matrix core and applies scale / zero to the fp32 group sums, i.e. it computes with the UNROUNDED weights.  The
# distance to the reference is therefore the reference's own weight rounding: measured 1.9e-4 .. 2.0e-4 (fp16) and 1.54e-3 .. 1.59e-3
# (bf16) of mean |y| on every case below, whatever the row count, group size, bit width or tile form — and the float64 oracle
# (tests/test_gpu_parity.py::test_rows5_*) says the HIP result is the closer one.  Bounds = 2.5 x the measured distance.
GOLD_R5 = os.path.join(ROOT, "tests", "golden", "fullsize_ref_r5.npz")
BOUNDS_R5 = {
    "cfgA_bf16_m16": 4e-3, "cfgA_fp16_m32": 5e-4, "cfgA_bf16_m32": 4e-3, "cfgA_fp16_m48": 5e-4, "cfgA_bf16_m48": 4e-3,
    "w4_g64_fp16_m32": 5e-4, "w4_g64_bf16_m32": 4e-3, "w4_g32_fp16_m32": 5e-4, "a16w2_4096_fp16_m32": 5e-4, "w4_8192x4096_fp16_m32": 5e-4,
    # decode kernels re-written in round 6 (gemv_mfma_kernel on counted asm loads, exact fp16 planes): the reference's GEMV family
    # accumulates in the 16-bit type (measured distance 4.7e-3 bf16 at 8192^2, 1.5e-3 fp16 2-bit at 4096^2)
    "cfgB_bf16_m1": 9e-3, "a16w2_4096_fp16_m1": 4e-3,
}


@pytest.mark.parametrize("name", sorted(BOUNDS_R5))
def test_hip_matches_reference_outputs_from_the_mi355x_round6_cases(name):
    import gemlite_amd
    from oracle.run_ref_gpu import COL0, COLSTEP
    gold = np.load(GOLD_R5)
    assert name in gold.files, f"{name} missing from the fixture"
    layer, x = _cases()[name](gemlite_amd)
    y = layer(x)
    torch.cuda.synchronize()
    dt = str(gold[name + "__dtype"])
    ref = torch.from_numpy(gold[name]).view({"torch.float16": torch.float16, "torch.bfloat16": torch.bfloat16}[dt]).float().numpy().astype(np.float64)
    got = y[:, COL0::COLSTEP].float().cpu().numpy().astype(np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all()
    rel = np.abs(got - ref).mean() / np.abs(ref).mean()
    assert rel < BOUNDS_R5[name], (name, rel, BOUNDS_R5[name])


def test_round6_fixture_is_the_reference_run_recorded_in_profiles():
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06", "reference_triton_mi355x_r5.json")))
    done = {r["case"] for r in rec["report"] if "us" in r and "error" not in r}
    assert set(BOUNDS_R5) <= done


# ---- round 6, second half: the territory of w8_rows_lds_kernel (gemm_w8_rows.hip) and group sizes that are not a power of two -------------------
# tests/golden/fullsize_ref_r6.npz = `python oracle/run_ref_gpu.py --which ref --only r6 --fixture fullsize_ref_r6.npz` on an MI355X (scripts/r6/run_u.sh;
# timings + the distances measured on the spot: profiles/r06/reference_triton_mi355x_r6.json, hip_same_method_r6.json).
#   * A8W8 (int8 and fp8 e4m3, per-token x per-channel scales) at 4 / 32 / 64 rows and the two-blocks-per-CU form at 8192^2: the distance measured was
#     EXACTLY 0 — int32 (fp32) accumulation of the same products and the reference's epilogue order, bit for bit;
#   * A16W8 (int8 / fp8 weights under fp16 / bf16): the reference rounds every scaled weight to the 16-bit type before tl.dot, the kernel converts the
#     8-bit code exactly and applies the channel scale once to the fp32 sum: 2.1e-4 (fp16) / 1.6e-3 (bf16) of mean |y|, the reference's own rounding;
#   * groups of 96 / 192 at 16 / 64 rows (the tile kernel, per-weight rounding like the reference): 2e-7 .. 4e-7; one row (dot-product GEMV against the
#     reference's fp16-accumulating GEMV): 1.4e-3.
GOLD_R6 = os.path.join(ROOT, "tests", "golden", "fullsize_ref_r6.npz")
BOUNDS_R6 = {
    **{f"a8w8_int8_m{m}": 1e-6 for m in (4, 32, 64)}, "a8w8_fp8_m32": 1e-6, "a8w8_fp8_m64": 1e-6, "a8w8_int8_8192_m8": 1e-6,
    **{f"a16w8_int8_fp16_m{m}": 5e-4 for m in (8, 32, 64)}, **{f"a16w8_int8_bf16_m{m}": 4e-3 for m in (8, 32, 64)},
    "a16w8_fp8_fp16_m32": 5e-4, "a16w8_int8_8192_fp16_m8": 5e-4,
    "w4_g96_fp16_m16": 1e-4, "w4_g192_bf16_m64": 1e-4, "w4_g96_fp16_m1": 4e-3,
}


@pytest.mark.parametrize("name", sorted(BOUNDS_R6))
def test_hip_matches_reference_outputs_from_the_mi355x_round6_second_half(name):
    import gemlite_amd
    from oracle.run_ref_gpu import COL0, COLSTEP
    gold = np.load(GOLD_R6)
    assert name in gold.files, f"{name} missing from the fixture"
    layer, x = _cases()[name](gemlite_amd)
    y = layer(x)
    torch.cuda.synchronize()
    dt = str(gold[name + "__dtype"])
    ref = torch.from_numpy(gold[name]).view({"torch.float16": torch.float16, "torch.bfloat16": torch.bfloat16}[dt]).float().numpy().astype(np.float64)
    got = y[:, COL0::COLSTEP].float().cpu().numpy().astype(np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all()
    rel = np.abs(got - ref).mean() / np.abs(ref).mean()
    assert rel < BOUNDS_R6[name], (name, rel, BOUNDS_R6[name])


def test_round6_second_fixture_is_the_reference_run_recorded_in_profiles():
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06", "reference_triton_mi355x_r6.json")))
    done = {r["case"] for r in rec["report"] if "us" in r and "error" not in r}
    assert set(BOUNDS_R6) <= done
