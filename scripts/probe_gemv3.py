"""Round-3 GEMV probe: per-launch time inside a replayed hipGraph (the bench's timed region) for kernel variants of the decode
shapes, with an on-the-spot parity check of every variant against the round-2 kernel (tuning[3] & 16) and the float64 oracle.
    gpurun -- 'python scripts/probe_gemv3.py [workload ...]'   (run it under rocprofv3 --kernel-trace --stats for device durations)"""
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
CASES = {
    "a16w2_16384_m1": [(0, 0, 0, 512), (0, 0, 0, 1024)],
    "a16w4_16384_m1": [(0, 0, 0, 512), (0, 0, 0, 1024)],
    "a16w4_8192_m1": [(0, 0, 0, 512), (0, 0, 0, 1024)],
    "a16w2_8192_m1": [(0, 0, 0, 512), (0, 0, 0, 1024)],
    "a16w4_11008n_m1": [(0, 0, 0, 512), (0, 0, 0, 1024)],
    "a16w4_11008_m1": [(0, 0, 0, 512), (0, 0, 0, 1024)],
    "a16w4_4096_m1": [(0, 0, 0, 512), (0, 0, 0, 1024)],
    "a16w4_8192_m4": [(0, 0, 0, 512), (0, 0, 0, 0)],
}
bench.WORKLOADS.update({"a16w2_8192_m1": (8192, 8192, 2, 128, 1, "fp16", 16, "hbm"), "a16w2_4096_m1": (4096, 4096, 2, 128, 1, "fp16", 32, "hbm"), "a16w2_11008_m1": (11008, 4096, 2, 128, 1, "fp16", 24, "hbm"), "a16w4_11008n_m1": (11008, 4096, 4, 128, 1, "fp16", 12, "hbm")})
bench.WORKLOADS.update({"a16w4_4096_m2": (4096, 4096, 4, 128, 2, "fp16", 32, "hbm"), "a16w4_4096_m4": (4096, 4096, 4, 128, 4, "fp16", 32, "hbm"), "a16w4_8192_m4": (8192, 8192, 4, 128, 4, "fp16", 8, "hbm")})
only = [a for a in sys.argv[1:] if not a.startswith("-")]
bench.WORKLOADS.update({"a16w4_8192_m16": (8192, 8192, 4, 128, 16, "fp16", 8, "hbm"), "a16w4_4096_m32": (4096, 4096, 4, 128, 32, "fp16", 32, "hbm")})
for a in only:   # any bench workload can be named; it gets the default list unless --tunings is given
    if a not in CASES and a in bench.WORKLOADS:
        CASES[a] = [(0, 0, 0, 0)]
for a in sys.argv[1:]:   # --tunings='[[22,0,0,1024],[24,0,8,1024]]' replaces the list of every selected case
    if a.startswith("--tunings="):
        for k in CASES:
            CASES[k] = [tuple(t) for t in json.loads(a.split("=", 1)[1])]
for name, tunings in CASES.items():
    if only and name not in only:
        continue
    ref_out = None
    for t in tunings:
        core.TUNING_OVERRIDE = t
        try:
            r = bench.Runner(name, dev, lib)
            y = r.call(r.mods[0]).float().cpu().numpy()
            torch.cuda.synchronize()
            if ref_out is None:
                ref_out = y
            scale = np.abs(ref_out).mean()
            c_us, steps, el = r.chained_us_per_launch(min_seconds=0.2)
            print(json.dumps(dict(workload=name, tuning=t, kernel=r.kernel_name(), chained_us=round(c_us, 3),
                                  gbs=round(r.bytes / 1e3 / c_us, 1), frac=round(r.bytes / 1e3 / c_us / 8000, 4),
                                  rel_vs_first=float(np.abs(y - ref_out).mean() / scale), max_vs_first=float(np.abs(y - ref_out).max() / scale),
                                  bitwise_equal_first=bool(np.array_equal(y, ref_out)))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:160])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
