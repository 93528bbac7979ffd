"""Round 3: the K-slice combine of the 8-wave MFMA kernel — reduce-scatter between co-resident slices (default where it
applies) against slabs + ticket (tuning[3] & 128) — per-launch time inside a replayed hipGraph (>= 32 launches per graph),
outputs compared bitwise between the two protocols.    python scripts/probe_mma3.py [cfgA cfgB m64 m128 w2 ...]"""
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
bench.WORKLOADS.update({
    "a16w4_4096_m64": (4096, 4096, 4, 128, 64, "bf16", 32, "mfma"), "a16w4_4096_m128": (4096, 4096, 4, 128, 128, "bf16", 32, "mfma"),
    "a16w4_4096_m512": (4096, 4096, 4, 128, 512, "bf16", 32, "mfma"), "a16w4_4096_m1024": (4096, 4096, 4, 128, 1024, "bf16", 32, "mfma"),
    "a16w4_11008x4096_m256": (11008, 4096, 4, 128, 256, "bf16", 12, "mfma"), "a16w4_4096x14336_m256": (4096, 14336, 4, 128, 256, "bf16", 10, "mfma"),
})
T, H = 128, 256   # tuning[3]: 128 = slab + ticket combine, 256 = reduce-scatter with immediate hand-over (every block polls once)
CASES = {
    "cfgA": ("a16w4_4096_m256", [(0, 0, 0, 0), (0, 0, 0, T), (0, 0, 0, H), (0, 2, 2, 0), (0, 2, 2, T), (0, 4, 4, 0), (0, 4, 4, T), (0, 4, 4, H), (0, 2, 4, 0), (0, 2, 4, T),
                                 (0, 4, 8, 0), (0, 4, 8, T), (0, 8, 8, 0)]),
    "cfgB": ("a16w4_8192_m256", [(0, 0, 0, 0), (0, 0, 0, T), (0, 0, 0, H), (0, 2, 4, 0), (0, 2, 4, T), (0, 4, 4, 0), (0, 4, 8, 0), (0, 4, 8, T), (0, 2, 8, 0), (0, 2, 8, T)]),
    "m64": ("a16w4_4096_m64", [(0, 0, 0, 0), (0, 0, 0, T), (0, 2, 2, 0), (0, 2, 2, T)]),
    "m128": ("a16w4_4096_m128", [(0, 0, 0, 0), (0, 0, 0, T), (0, 2, 2, 0), (0, 2, 4, 0), (0, 4, 4, 0), (0, 4, 4, T)]),
    "m512": ("a16w4_4096_m512", [(0, 0, 0, 0), (0, 0, 0, T), (0, 2, 4, 0), (0, 2, 8, 0), (0, 4, 8, 0), (0, 1, 4, 0)]),
    "m1024": ("a16w4_4096_m1024", [(0, 0, 0, 0), (0, 0, 0, T), (0, 2, 8, 0), (0, 1, 8, 0)]),
    "w2": ("a16w2_16384_m256", [(0, 0, 0, 0), (0, 0, 0, T), (0, 4, 8, 0)]),
    "n11008": ("a16w4_11008x4096_m256", [(0, 0, 0, 0), (0, 0, 0, T), (0, 2, 8, 0), (0, 2, 4, 0)]),
    "k14336": ("a16w4_4096x14336_m256", [(0, 0, 0, 0), (0, 0, 0, T), (0, 8, 8, 0), (0, 4, 4, 0), (0, 4, 8, 0)]),
    "pre": ("a16w4_8192_m2048", [(0, 0, 0, 0), (0, 1, 8, 0)]),
}
for key in (sys.argv[1:] or ["cfgA", "cfgB"]):
    name, tunings = CASES[key]
    outs = {}
    for t in tunings:
        core.TUNING_OVERRIDE = t
        try:
            r = bench.Runner(name, dev, lib)
            y = r.call(r.mods[0]).float().cpu().numpy()
            torch.cuda.synchronize()
            kn = r.kernel_name()
            c_us, n, el = r.chained_us_per_launch(min_seconds=0.2)
            tw = (t[0], t[1], t[2], 0)
            same = None
            if tw in outs:
                same = bool(np.array_equal(outs[tw], y))
            outs.setdefault(tw, y)
            print(json.dumps(dict(workload=name, tuning=t, kernel=kn, combine="ticket" if t[3] & T else ("hand-over" if t[3] & H else "auto"), chained_us=round(c_us, 3),
                                  tflops=round(r.flops / c_us / 1e6, 1), frac=round(r.flops / c_us / 1e6 / 2500, 4),
                                  bitwise_equal_first_of_same_tile=same, finite=bool(np.isfinite(y).all()))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, tuning=t, error=f"{type(e).__name__}: {e}"[:200])), flush=True)
        finally:
            core.TUNING_OVERRIDE = None
        torch.cuda.empty_cache()
