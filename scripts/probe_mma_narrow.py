"""Round 4: narrow (32 MI x 64, K unsplit) tiles of the W4 MFMA kernel (tuning[2] = 32 + variant) against the library's choice;
graph-replayed time per launch + error against a float64 reference of the first layer.    python scripts/probe_mma_narrow.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
W = bench.WORKLOADS
for m in (96, 128, 192, 384, 512):
    W[f"a16w4_4096_m{m}"] = (4096, 4096, 4, 128, m, "bf16", 64, "mfma")
W["a16w4_8192_m128"] = (8192, 8192, 4, 128, 128, "bf16", 16, "mfma")
W["a16w4_8192_m512"] = (8192, 8192, 4, 128, 512, "bf16", 16, "mfma")
W["a16w4_11008x4096_m256"] = (11008, 4096, 4, 128, 256, "bf16", 24, "mfma")
W["a16w4_4096x11008_m256"] = (4096, 11008, 4, 128, 256, "bf16", 24, "mfma")
W["a16w4_4096_m256_fp16"] = (4096, 4096, 4, 128, 256, "fp16", 64, "mfma")
D = (0, 0, 0, 0)
CASES = {
    "a16w4_4096_m256": [D, D, (0, 0, 32, 0), (0, 0, 33, 0), (0, 2, 34, 0), D, (0, 0, 32, 0)],
    "a16w4_4096_m256_fp16": [D, D, (0, 0, 32, 0), (0, 0, 33, 0)],
    "a16w4_4096_m128": [D, D, (0, 0, 32, 0), (0, 2, 32, 0), (0, 2, 33, 0), (0, 4, 32, 0), D],
    "a16w4_4096_m96": [D, D, (0, 0, 32, 0), (0, 2, 32, 0), (0, 4, 32, 0), D],
    "a16w4_4096_m192": [D, D, (0, 0, 32, 0), (0, 2, 32, 0), (0, 2, 34, 0), D],
    "a16w4_4096_m384": [D, D, (0, 0, 32, 0), (0, 0, 34, 0), D],
    "a16w4_4096_m512": [D, D, (0, 0, 34, 0), (0, 0, 32, 0), D],
    "a16w4_8192_m128": [D, D, (0, 0, 32, 0), (0, 0, 33, 0), (0, 0, 34, 0), D],
    "a16w4_8192_m256": [D, D, (0, 0, 34, 0), D],
    "a16w4_8192_m512": [D, D, (0, 0, 34, 0), D],
    "a16w4_11008x4096_m256": [D, D, (0, 0, 32, 0), (0, 0, 34, 0), D],
    "a16w4_4096x11008_m256": [D, D, (0, 0, 32, 0), (0, 0, 33, 0), (0, 0, 34, 0), D],
    "a16w4_4096x14336_m256": [D, D, (0, 0, 32, 0), (0, 0, 33, 0), D],
    "a16w4_14336x4096_m256": [D, D, (0, 0, 32, 0), (0, 0, 34, 0), D],
    "a16w4_5120_m256": [D, D, (0, 0, 32, 0), (0, 0, 34, 0), D],
    "a16w2_4096_m256": [D, D, (0, 0, 32, 0), D],
}
W["a16w4_4096x14336_m256"] = (4096, 14336, 4, 128, 256, "bf16", 16, "mfma")
W["a16w4_14336x4096_m256"] = (14336, 4096, 4, 128, 256, "bf16", 16, "mfma")
W["a16w4_5120_m256"] = (5120, 5120, 4, 128, 256, "bf16", 40, "mfma")
W["a16w2_4096_m256"] = (4096, 4096, 2, 128, 256, "bf16", 64, "mfma")
W["a16w4_4096_m192"] = (4096, 4096, 4, 128, 192, "bf16", 64, "mfma")
only = sys.argv[1:]
for name, tunings in CASES.items():
    if only and name not in only:
        continue
    N, K, nbits, group, M, dt, _, _ = W[name]
    ref = None
    for t in tunings:
        core.TUNING_OVERRIDE = t
        try:
            r = bench.Runner(name, dev, lib)
            lin = r.mods[0]
            y = r.call(lin).float().cpu().numpy()
            torch.cuda.synchronize()
            if ref is None and N * K <= 8192 * 8192:  # float64 reference from the packed tensors (fma mode: zeros already folded)
                e = 32 // nbits
                P = lin.W_q.cpu().numpy().astype(np.uint32)
                codes = np.zeros((P.shape[0] * e, N), dtype=np.float32)
                for i in range(e):
                    codes[i::e] = (P >> (nbits * i)) & ((1 << nbits) - 1)
                s_ = lin.scales.float().cpu().numpy()
                z_ = lin.zeros.float().cpu().numpy()
                Wd = (codes * np.repeat(s_, group, axis=0) + np.repeat(z_, group, axis=0)).astype(np.float64)
                ref = r.x.float().cpu().numpy().astype(np.float64) @ Wd
            rel = float(np.abs(y - ref).mean() / np.abs(ref).mean()) if ref is not None else None
            c_us, n, el = r.chained_us_per_launch(min_seconds=0.2)
            print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3), tflops=round(r.flops / c_us / 1e6, 1),
                                  frac=round(r.flops / c_us / 1e6 / 2500, 4), rel_err_vs_f64=rel)), flush=True)
            del r
        except Exception as ex:
            print(json.dumps(dict(workload=name, tuning=t, error=f"{type(ex).__name__}: {ex}"[:200])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
