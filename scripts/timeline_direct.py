"""Round 3: per-block timeline of the few-row MFMA kernel (gemm_wn_direct.hip).  Needs a development build:
    make -C gemlite_amd/csrc DIRECT_EXTRA=-DGL_DIRECT_TIMELINE
Stamps (100 MHz global clock, wave 0 of every block): 0 start | 1 both pieces requested | 2 first piece consumed | 3 all pieces
consumed | 4 waves joined in LDS | 5 output stored / partial sent | 6 ticket says last | 7 combined output stored.
The launch is the last node of a replayed hipGraph (back-to-back launches).    python scripts/timeline_direct.py [workload] [tunings-json]"""
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
name = sys.argv[1] if len(sys.argv) > 1 else "a16w4_4096_m16"
tunings = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [[0, 0, 0, 0], [2, 2, 8, 0], [2, 1, 8, 0]]
for tun in tunings:
    tun = (tun[0], tun[1], tun[2], tun[3] | 4)
    core.TUNING_OVERRIDE = tun
    r = bench.Runner(name, dev, lib)
    c_us, _, _ = r.chained_us_per_launch(min_seconds=0.1)
    recs = []
    for rep in range(8):
        r.run_step()
        torch.cuda.synchronize()
        ws = [w for w in _hip._workspaces.values()][-1]
        st = ws[(65536 - 4096) * 4: (65536 - 4096) * 4 + 512 * 8 * 8].view(torch.int64).cpu().numpy().reshape(512, 8).astype(np.float64)
        recs.append(st.copy())
    core.TUNING_OVERRIDE = None
    out = {"workload": name, "tuning": tun, "kernel": r.kernel_name(), "chained_us": round(c_us, 3)}
    acc = {}
    for st in recs[2:]:
        live = st[:, 0] > 0
        st = st[live]
        t0 = st[:, 0].min()
        rel = (st - t0) / 100.0
        out["blocks"] = int(live.sum())
        for i, nm in ((0, "start"), (1, "requested"), (2, "first_piece"), (3, "loop_done"), (4, "joined"), (5, "stored_or_sent"), (7, "combined_stored")):
            col = rel[:, i]
            col = col[col > -1e6]
            if i == 7:
                col = col[st[:len(col), 7] > st[:len(col), 0]] if False else rel[:, 7][st[:, 7] >= st[:, 0]]
            if len(col):
                acc.setdefault(nm, []).append(np.percentile(col, [10, 50, 90, 100]))
    for k, v in acc.items():
        out[k + "_p10_p50_p90_max"] = np.mean(v, axis=0).round(2).tolist()
    print(json.dumps(out), flush=True)
    del r
    torch.cuda.empty_cache()
