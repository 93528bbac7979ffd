#!/usr/bin/env python3
"""Run THE REFERENCE (its Triton kernels) on the MI355X next to the HIP library.  TEST INFRASTRUCTURE — never imported by
the product.

`bash oracle/make_ref.sh` (build container) stages /root/reference/gemlite under oracle/_ref/ (git-ignored, shipped by
`gpurun`); this script runs on the GPU box and
  (a) dumps the reference's outputs at BASELINE sizes — cfgA M in {1, 16, 256}, cfgB M = 256 bf16, A8W8 int8 / fp8, A16W2
      16384^2, the block-scaled processors (tl.dot_scaled runs natively here) — as column-subsampled golden fixtures
      (`fullsize_ref.npz`: every 16th column, all rows; inputs are regenerated from seeds by the test), and compares the
      HIP library with them on the spot;
  (b) times reference and HIP with the reference's own method (examples/benchmark_triton.py:44-60: 256 MiB cache flush before
      every call, one event pair per call, min and median) -> `reference_triton_mi355x.json`.
Outputs land in gpurun_out/ref/; the builder copies the fixture to tests/golden/ and the timings to profiles/.

Usage (GPU box):  python oracle/run_ref_gpu.py --which ref [--budget-s 480]; python oracle/run_ref_gpu.py --which hip
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gemlite_oracle as O  # noqa: E402

DEV = "cuda:0"
COL0, COLSTEP = 5, 16  # golden subsample: columns 5, 21, 37, ...


def import_reference():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import gemlite  # the staged reference

    assert "oracle/_ref" in gemlite.__file__.replace("\\", "/"), gemlite.__file__
    sys.path.pop(0)
    return gemlite


def bits(t):
    t = t.detach().cpu().contiguous()
    return t.view(torch.int16).numpy() if t.dtype in (torch.float16, torch.bfloat16) else t.numpy()


# ---------------------------------------------------------------------------------------------------------------- cases
# every builder takes the package (reference or gemlite_amd: same API) and returns (layer, x)
def case_int(N, K, nbits, gs, tdt, M, seed, xseed):
    def build(pkg):
        W_q, scales, zeros = O.gen_data(N, K, nbits, gs, seed=seed)
        code = {torch.float16: pkg.DType.FP16, torch.bfloat16: pkg.DType.BF16}[tdt]
        lin = pkg.GemLiteLinear(nbits, gs, K, N, code, code)
        lin.pack(torch.from_numpy(W_q).to(DEV), torch.from_numpy(scales.astype(np.float32)).to(tdt).to(DEV),
                 torch.from_numpy(zeros.astype(np.float32)).to(tdt).to(DEV), None)
        x = torch.from_numpy(O.gen_x(M, K, seed=xseed).astype(np.float32)).to(tdt).to(DEV)
        return lin, x
    return build


def case_helper(proc_name, N, K, M, seed, tdt=torch.float16, from_linear=False, **kw):
    """kw: constructor arguments.  fp8=torch.float8_e4m3fn pins the OCP format: the reference's default on ROCm is the MI300X
    fnuz format (helper.py:13-19), which gfx950 only emulates (first run: 2.7 ms for FP8 x FP8 16384^2 M = 256)."""
    def build(pkg):
        torch.manual_seed(seed)
        W = (torch.randn(N, K) / 30).to(tdt)
        proc = getattr(pkg.helper, proc_name)(device=DEV, dtype=tdt, **kw)
        if from_linear:
            lin0 = torch.nn.Linear(K, N, bias=False, dtype=tdt)
            lin0.weight.data = W
            layer = proc.from_linear(lin0.to(DEV), del_orig=False)
        else:
            layer = proc.from_weights(W)
        x = (torch.randn(M, K) / 10).to(tdt).to(DEV)
        return layer, x
    return build


def case_a8wn(N, K, nbits, gs, M, seed, xseed):
    """A8Wn_HQQ_INT_dynamic (helper.py:502-615): 4-bit grouped weights under DYNAMICALLY quantised fp8 e4m3 activations."""
    def build(pkg):
        W_q, scales, zeros = O.gen_data(N, K, nbits, gs, seed=seed)
        proc = pkg.helper.A8Wn_HQQ_INT_dynamic(device=DEV, dtype=torch.float16, W_nbits=nbits, fp8=torch.float8_e4m3fn)
        layer = proc.from_weights(torch.from_numpy(W_q).to(DEV), torch.from_numpy(scales.astype(np.float32)).half().to(DEV),
                                  torch.from_numpy(zeros.astype(np.float32)).half().to(DEV))
        x = torch.from_numpy(O.gen_x(M, K, seed=xseed).astype(np.float32)).half().to(DEV)
        return layer, x
    return build


# Round 4: the kernels added since the first fixture had no reference-output pin — bf16 decode, the 2 .. 8-row
# MFMA GEMV / few-row kernels, 64 rows, 2-bit bf16 decode, fp8-dynamic activations over packed weights.  Written to
# fullsize_ref_r4.npz by `--which ref --only <these>`; the first fixture is left as recorded.
CASES_R4 = (
    [("cfgA_bf16_m1", case_int(4096, 4096, 4, 128, torch.bfloat16, 1, 0, 1), (1, 4096, 4096))]
    + [(f"cfgA_{tn}_m{m}", case_int(4096, 4096, 4, 128, tdt, m, 0, m), (m, 4096, 4096))
       for m in (2, 4, 8, 64) for tn, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16))]
    + [("a16w2_16384_bf16_m1", case_int(16384, 16384, 2, 128, torch.bfloat16, 1, 5, 2), (1, 16384, 16384)),
       ("a8w4_fp8dyn_m1", case_a8wn(4096, 4096, 4, 128, 1, 11, 3), (1, 4096, 4096)),
       ("a8w4_fp8dyn_m16", case_a8wn(4096, 4096, 4, 128, 16, 11, 4), (16, 4096, 4096)),
       ("a8w4_fp8dyn_m256", case_a8wn(4096, 4096, 4, 128, 256, 11, 5), (256, 4096, 4096))])

# Round 6: the territory of gemm_w4_rows_kernel (round 5) had reference outputs at M = 64 g128 only.  16 / 32 / 48 rows in
# both types, groups of 64 and 32 at 32 rows, 2-bit words at 32 rows, and the two-column-tile form (4096 < N <= 8192).  Written to
# fullsize_ref_r5.npz by `--which ref --only r5 --fixture fullsize_ref_r5.npz`.
CASES_R5 = (
    [("cfgA_bf16_m16", case_int(4096, 4096, 4, 128, torch.bfloat16, 16, 0, 16), (16, 4096, 4096))]
    + [(f"cfgA_{tn}_m{m}", case_int(4096, 4096, 4, 128, tdt, m, 0, m), (m, 4096, 4096))
       for m in (32, 48) for tn, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16))]
    + [("w4_g64_fp16_m32", case_int(4096, 4096, 4, 64, torch.float16, 32, 21, 22), (32, 4096, 4096)),
       ("w4_g64_bf16_m32", case_int(4096, 4096, 4, 64, torch.bfloat16, 32, 21, 22), (32, 4096, 4096)),
       ("w4_g32_fp16_m32", case_int(4096, 4096, 4, 32, torch.float16, 32, 23, 24), (32, 4096, 4096)),
       ("a16w2_4096_fp16_m32", case_int(4096, 4096, 2, 128, torch.float16, 32, 25, 26), (32, 4096, 4096)),
       ("w4_8192x4096_fp16_m32", case_int(8192, 4096, 4, 128, torch.float16, 32, 27, 28), (32, 8192, 4096)),
       # (round 6) the decode kernels this round re-wrote, pinned as well: 8192^2 bf16 M = 1 and the 2-bit 4096^2 M = 1
       ("cfgB_bf16_m1", case_int(8192, 8192, 4, 128, torch.bfloat16, 1, 3, 7), (1, 8192, 8192)),
       ("a16w2_4096_fp16_m1", case_int(4096, 4096, 2, 128, torch.float16, 1, 25, 29), (1, 4096, 4096))])

# Round 6, second half: the territory of w8_rows_lds_kernel (gemm_w8_rows.hip: unpacked 8-bit weights at 2 .. 64 rows, x through LDS) and group sizes that
# are not a power of two (tile kernel, gs_magic).  Written to fullsize_ref_r6.npz by `--which ref --only r6 --fixture fullsize_ref_r6.npz`.
CASES_R6 = (
    [(f"a8w8_int8_m{m}", case_helper("A8W8_int8_dynamic", 4096, 4096, m, 100 + m), (m, 4096, 4096)) for m in (4, 32, 64)]
    + [(f"a8w8_fp8_m{m}", case_helper("A8W8_fp8_dynamic", 4096, 4096, m, 120 + m, fp8=torch.float8_e4m3fn), (m, 4096, 4096)) for m in (32, 64)]
    + [(f"a16w8_int8_{tn}_m{m}", case_helper("A16W8_INT8", 4096, 4096, m, 140 + m, tdt), (m, 4096, 4096))
       for m in (8, 32, 64) for tn, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16))]
    + [("a16w8_fp8_fp16_m32", case_helper("A16W8_FP8", 4096, 4096, 32, 171, fp8=torch.float8_e4m3fn), (32, 4096, 4096)),
       ("a8w8_int8_8192_m8", case_helper("A8W8_int8_dynamic", 8192, 8192, 8, 181), (8, 8192, 8192)),      # the two-blocks-per-CU form
       ("a16w8_int8_8192_fp16_m8", case_helper("A16W8_INT8", 8192, 8192, 8, 182), (8, 8192, 8192)),
       ("w4_g96_fp16_m16", case_int(4096, 3072, 4, 96, torch.float16, 16, 31, 32), (16, 4096, 3072)),      # odd multiple of 32: two metadata pairs per sub-block
       ("w4_g192_bf16_m64", case_int(4096, 3072, 4, 192, torch.bfloat16, 64, 33, 34), (64, 4096, 3072)),
       ("w4_g96_fp16_m1", case_int(4096, 3072, 4, 96, torch.float16, 1, 31, 35), (1, 4096, 3072))])

CASES = [
    # name, builder, shape-for-flops
    ("cfgA_fp16_m1", case_int(4096, 4096, 4, 128, torch.float16, 1, 0, 1), (1, 4096, 4096)),
    ("cfgA_fp16_m16", case_int(4096, 4096, 4, 128, torch.float16, 16, 0, 16), (16, 4096, 4096)),
    ("cfgA_fp16_m256", case_int(4096, 4096, 4, 128, torch.float16, 256, 0, 256), (256, 4096, 4096)),
    ("cfgA_bf16_m256", case_int(4096, 4096, 4, 128, torch.bfloat16, 256, 0, 256), (256, 4096, 4096)),
    ("cfgB_bf16_m256", case_int(8192, 8192, 4, 128, torch.bfloat16, 256, 3, 7), (256, 8192, 8192)),
    ("cfgB_fp16_m1", case_int(8192, 8192, 4, 128, torch.float16, 1, 3, 7), (1, 8192, 8192)),
    ("a8w8_int8_m1", case_helper("A8W8_int8_dynamic", 4096, 4096, 1, 1), (1, 4096, 4096)),
    ("a8w8_int8_m16", case_helper("A8W8_int8_dynamic", 4096, 4096, 16, 16), (16, 4096, 4096)),
    ("a8w8_int8_m256", case_helper("A8W8_int8_dynamic", 4096, 4096, 256, 256), (256, 4096, 4096)),
    ("a8w8_fp8_m16", case_helper("A8W8_fp8_dynamic", 4096, 4096, 16, 21, fp8=torch.float8_e4m3fn), (16, 4096, 4096)),
    ("a8w8_fp8_m256", case_helper("A8W8_fp8_dynamic", 4096, 4096, 256, 261, fp8=torch.float8_e4m3fn), (256, 4096, 4096)),
    ("a8w8_fp8_m1", case_helper("A8W8_fp8_dynamic", 4096, 4096, 1, 20, fp8=torch.float8_e4m3fn), (1, 4096, 4096)),
    ("a8w8_fnuz_default_m256", case_helper("A8W8_fp8_dynamic", 4096, 4096, 256, 261), (256, 4096, 4096)),  # reference only: its ROCm default format
    ("a16w2_16384_m1", case_int(16384, 16384, 2, 128, torch.float16, 1, 5, 2), (1, 16384, 16384)),
    ("a16w2_16384_m256", case_int(16384, 16384, 2, 128, torch.float16, 256, 5, 9), (256, 16384, 16384)),
    ("fp8_16384_m256", case_helper("A8W8_fp8_dynamic", 16384, 16384, 256, 77, fp8=torch.float8_e4m3fn), (256, 16384, 16384)),
    # block-scaled processors: the reference's own acceptance layer (tests/test_mxfp.py: 4096 -> 2048)
    ("mx_a8w8_m16", case_helper("A8W8_MXFP_dynamic", 2048, 4096, 16, 31, torch.bfloat16, True, fp8=torch.float8_e4m3fn), (16, 2048, 4096)),
    ("mx_a8w8_m256", case_helper("A8W8_MXFP_dynamic", 2048, 4096, 256, 32, torch.bfloat16, True, fp8=torch.float8_e4m3fn), (256, 2048, 4096)),
    ("mx_a8w4_m16", case_helper("A8W4_MXFP_dynamic", 2048, 4096, 16, 33, torch.bfloat16, True, fp8=torch.float8_e4m3fn), (16, 2048, 4096)),
    ("mx_a4w4_m16", case_helper("A4W4_MXFP_dynamic", 2048, 4096, 16, 34, torch.bfloat16, True), (16, 2048, 4096)),
    ("mx_a4w4_m256", case_helper("A4W4_MXFP_dynamic", 2048, 4096, 256, 35, torch.bfloat16, True), (256, 2048, 4096)),
    ("mx_a16w4_m16", case_helper("A16W4_MXFP", 2048, 4096, 16, 36, torch.bfloat16, True), (16, 2048, 4096)),
    ("nvfp4_m16", case_helper("A4W4_NVFP_dynamic", 2048, 4096, 16, 37, torch.bfloat16, True), (16, 2048, 4096)),
] + list(CASES_R4) + list(CASES_R5) + list(CASES_R6)

_FLUSH = None


def eval_time(fn, rep=100):
    """benchmark_triton.py:44-60: flush 256 MiB, one event pair per call."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.int, device=DEV)
    ts = []
    for i in range(rep):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _FLUSH.zero_()
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
        _FLUSH.add_(i & 7)
    ts = np.asarray(ts)
    return {"min_us": float(ts.min()), "median_us": float(np.median(ts[rep // 2:]))}


def graph_time(fn, reps=20, inner=16):
    """device time per call inside a replayed hipGraph (back-to-back, weights L2/MALL-warm): lower bound"""
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(inner):
                    fn()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * inner) * 1e6
    except Exception as ex:  # noqa: BLE001
        return f"graph capture failed: {type(ex).__name__}: {ex}"[:200]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", choices=["ref", "hip"], required=True,
                    help="both packages register the custom op gemlite::forward_functional, so they run in separate processes: "
                         "`ref` first (dumps full outputs to --tmp), then `hip` (compares with them)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref"))
    ap.add_argument("--tmp", default="/tmp/gemlite_ref_full")
    ap.add_argument("--budget-s", type=float, default=480.0)
    ap.add_argument("--fast-shapes", default="cfgA_fp16_m1,cfgA_bf16_m256,cfgB_bf16_m256")
    ap.add_argument("--only", default="", help="comma-separated case names, or `r4` / `r5` / `r6` = the additions of round 4 / round 6 (first, second half)")
    ap.add_argument("--fixture", default="fullsize_ref.npz", help="file name of the golden fixture written by --which ref")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    os.makedirs(args.tmp, exist_ok=True)
    t_start = time.time()
    pkg = import_reference() if args.which == "ref" else __import__("gemlite_amd")
    info = {"device": torch.cuda.get_device_properties(0).name, "torch": torch.__version__, "which": args.which,
            "triton": __import__("triton").__version__, "method": "256 MiB flush + event pair per call (benchmark_triton.py:44-60)"}
    golden, report = {}, []
    rounds = {"r4": CASES_R4, "r5": CASES_R5, "r6": CASES_R6}
    only = set(c[0] for c in rounds[args.only]) if args.only in rounds else set(filter(None, args.only.split(",")))
    tag = "_" + args.only if args.only in rounds else ""
    out_json = os.path.join(args.out, f"reference_triton_mi355x{tag}.json" if args.which == "ref" else f"hip_same_method{tag}.json")

    def run_phase(mode, names, dump):
        if args.which == "ref":
            pkg.set_autotune(mode)
        for name, build, shape in CASES:
            if name not in names or (only and name not in only):
                continue
            if time.time() - t_start > args.budget_s:
                report.append({"case": name, "mode": mode, "skipped": "time budget"})
                continue
            rec = {"case": name, "mode": mode, "shape_MNK": shape}
            try:
                t0 = time.time()
                lin, x = build(pkg)
                y = lin(x)
                torch.cuda.synchronize()
                rec["first_call_s"] = round(time.time() - t0, 2)
                rec["finite"] = bool(torch.isfinite(y.float()).all().item())
                full = os.path.join(args.tmp, name + ".pt")
                if args.which == "ref":
                    if dump:
                        torch.save(y.cpu(), full)
                        golden[name] = bits(y[:, COL0::COLSTEP])
                        golden[name + "__dtype"] = np.array(str(y.dtype))
                elif os.path.exists(full):
                    yr = torch.load(full).float().numpy().astype(np.float64)
                    yh = y.float().cpu().numpy().astype(np.float64)
                    scale = max(float(np.abs(yr).mean()), 1e-12)
                    rec.update(rel_mean_hip_vs_ref=float(np.abs(yh - yr).mean() / scale),
                               rel_max_hip_vs_ref=float(np.abs(yh - yr).max() / scale), mean_abs_ref=scale)
                rec["us"] = eval_time(lambda: lin(x))
                rec["graph_us"] = graph_time(lambda: lin(x))
            except Exception as ex:  # noqa: BLE001
                rec["error"] = f"{type(ex).__name__}: {ex}"[:400]
            report.append(rec)
            print(json.dumps(rec), flush=True)
            lin = x = y = None
            torch.cuda.empty_cache()
            json.dump({"info": info, "report": report}, open(out_json, "w"), indent=1)

    run_phase("default", [c[0] for c in CASES], dump=True)
    if args.which == "ref":
        np.savez_compressed(os.path.join(args.out, args.fixture), col0=np.array(COL0), colstep=np.array(COLSTEP), **golden)
        if not only:
            run_phase("fast", [n for n in args.fast_shapes.split(",") if n], dump=False)
    info["elapsed_s"] = round(time.time() - t_start, 1)
    json.dump({"info": info, "report": report}, open(out_json, "w"), indent=1)
    print("done in", info["elapsed_s"], "s")


if __name__ == "__main__":
    main()
