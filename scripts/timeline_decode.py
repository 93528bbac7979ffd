"""Round 3: when do the blocks of the decode GEMV start and finish?  tuning[3] & 4 makes wave 0 of every block store the
100 MHz global clock at [start, requests issued, arithmetic done, output stored]; the launch runs as the LAST node of a
replayed hipGraph of 32 distinct layers (the bench's timed region), so the stamps are those of a back-to-back launch.
    python scripts/timeline_decode.py [workload]"""
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
name = sys.argv[1] if len(sys.argv) > 1 else "a16w4_4096_m1"
for tun in ((0, 0, 0, 4 | 4096), (0, 0, 0, 4), (0, 0, 0, 4 | 4096), (0, 0, 0, 4)):  # round-3 decode kernel vs decode3
    core.TUNING_OVERRIDE = tun
    r = bench.Runner(name, dev, lib)
    c_us, _, _ = r.chained_us_per_launch(min_seconds=0.1)
    recs = []
    for rep in range(8):
        r.run_step()
        torch.cuda.synchronize()
        ws = [w for w in _hip._workspaces.values()][-1]
        nb = min(1024, r.N // 16)
        st = ws[(65536 - 4096) * 4: (65536 - 4096) * 4 + nb * 4 * 8].view(torch.int64).cpu().numpy().reshape(nb, 4).astype(np.float64)
        recs.append(st)
    core.TUNING_OVERRIDE = None
    out = {"workload": name, "tuning": tun, "kernel": r.kernel_name(), "chained_us": round(c_us, 3)}
    sp, st0, iss, ari, dur = [], [], [], [], []
    for st in recs[2:]:
        t0 = st[:, 0].min()
        rel = (st - t0) / 100.0  # us
        sp.append(rel[:, 3].max())
        st0.append(np.percentile(rel[:, 0], [50, 90, 100]))
        iss.append(np.percentile(rel[:, 1] - rel[:, 0], [50, 100]))
        ari.append(np.percentile(rel[:, 2] - rel[:, 0], [10, 50, 90, 100]))
        dur.append(np.percentile(rel[:, 3] - rel[:, 0], [10, 50, 90, 100]))
    out["span_first_start_to_last_end_us"] = round(float(np.mean(sp)), 3)
    out["block_start_p50_p90_max_us"] = np.mean(st0, axis=0).round(3).tolist()
    out["issue_p50_max_us"] = np.mean(iss, axis=0).round(3).tolist()
    out["start_to_arith_done_p10_p50_p90_max_us"] = np.mean(ari, axis=0).round(3).tolist()
    out["block_lifetime_p10_p50_p90_max_us"] = np.mean(dur, axis=0).round(3).tolist()
    print(json.dumps(out), flush=True)
    del r
    torch.cuda.empty_cache()
