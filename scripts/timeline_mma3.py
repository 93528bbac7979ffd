"""Round 3: per-block global-clock timeline of the 8-wave MFMA kernel as the last launch of a replayed hipGraph (tuning[3] & 4).
Stamps per block (wave 0): 0 start | 1 prologue barrier | 2 first step | 3 loop done | 4 K halves added | 5 partial tile sent |
6 peers arrived / ticket | 7 output stored.    python scripts/timeline_mma3.py [cfgA cfgB]"""
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
T = 128
CASES = {"cfgA": ("a16w4_4096_m256", [(0, 0, 0, 0), (0, 2, 2, T), (0, 4, 4, 0), (0, 4, 8, 0)]),
         "cfgB": ("a16w4_8192_m256", [(0, 0, 0, 0), (0, 2, 4, T), (0, 4, 8, 0), (0, 4, 8, T)])}
CASES["cfgA_r4"] = ("a16w4_4096_m256", [(0, 0, 0, 0), (0, 0, 33, 0), (0, 0, 0, 16384)])   # round 4: narrow tiles (64 x 64, K unsplit) vs the round-3 choice
for key in (sys.argv[1:] or ["cfgA", "cfgB"]):
    name, tunings = CASES[key]
    for tun in tunings:
        core.TUNING_OVERRIDE = (tun[0], tun[1], tun[2], tun[3] | 4)
        r = bench.Runner(name, dev, lib)
        kn = r.kernel_name()
        c_us, _, _ = r.chained_us_per_launch(min_seconds=0.1)
        recs = []
        for rep in range(8):
            r.run_step()
            torch.cuda.synchronize()
            ws = [w for w in _hip._workspaces.values()][-1]
            st = ws[(65536 - 4096) * 4: (65536 - 4096) * 4 + 256 * 8 * 8].view(torch.int64).cpu().numpy().reshape(256, 8).astype(np.float64)
            recs.append(st)
        core.TUNING_OVERRIDE = None
        out = {"workload": name, "tuning": tun, "kernel": kn, "chained_us": round(c_us, 3)}
        acc = {k: [] for k in ("span", "start", "s1", "s2", "s3", "s4", "s5", "s6", "s7")}
        for st in recs[2:]:
            live = st[:, 0] > 0
            st = st[live]
            t0 = st[:, 0].min()
            rel = (st - t0) / 100.0
            done = rel[:, 7] > 0  # blocks that reached the output stage (ticket protocol: only the last arriver)
            acc["span"].append(max(rel[:, 6].max(), rel[done, 7].max() if done.any() else 0))
            acc["start"].append(np.percentile(rel[:, 0], [50, 100]))
            for i in range(1, 7):
                acc[f"s{i}"].append(np.percentile(rel[:, i] - rel[:, 0], [10, 50, 90, 100]))
            if done.any():
                acc["s7"].append(np.percentile(rel[done, 7] - rel[done, 0], [10, 50, 90, 100]))
        out["blocks"] = int(live.sum())
        out["span_us"] = round(float(np.mean(acc["span"])), 2)
        out["block_start_p50_max"] = np.mean(acc["start"], axis=0).round(2).tolist()
        for i, nm in ((1, "prologue"), (2, "first_step"), (3, "loop_done"), (4, "khalves"), (5, "sent"), (6, "arrived"), (7, "stored")):
            if acc[f"s{i}"]:
                out[f"t_{nm}_p10_p50_p90_max"] = np.mean(acc[f"s{i}"], axis=0).round(2).tolist()
        print(json.dumps(out), flush=True)
        del r
        torch.cuda.empty_cache()
