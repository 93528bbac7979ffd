"""Development: s_memtime timeline of the GEMV kernel (block 0) — python scripts/timeline.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gemlite_amd
from gemlite_amd import GemLiteLinear, DType, _hip
from gemlite_amd.core import _hip_matmul
from oracle import gemlite_oracle as O
dev = "cuda:0"
out = {}
for name, (N, K, tun) in {"cfgA_tile16_xd_4w": (4096, 4096, (2, 1, 4, 4 | 2)), "cfgA_tile16_xd_8w": (4096, 4096, (2, 1, 8, 4 | 2)),
                          "cfgA_tile16_xd_16w": (4096, 4096, (2, 1, 16, 4 | 2))}.items():
    layers = []
    for i in range(24 if N == 4096 else 8):
        W_q, s, z = O.gen_data(N, K, 4, 128, seed=i)
        lin = GemLiteLinear(4, 128, K, N, DType.FP16, DType.FP16)
        lin.pack(torch.from_numpy(W_q).to(dev), torch.from_numpy(s).to(dev), torch.from_numpy(z).to(dev))
        layers.append(lin)
    x = torch.from_numpy(O.gen_x(1, K, seed=1)).to(dev)
    recs = []
    for rep in range(3):
        for lin in layers:   # rotate layers: cache-cold like the bench
            y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), 1, tun)
            torch.cuda.synchronize()
            ws = list(_hip._workspaces.values())[0]
            nw = tun[2] if tun[2] in (8, 16) else 4
            st = ws[(65536 - 4096) * 4: (65536 - 4096) * 4 + nw * 16 * 8].view(torch.int64).cpu().numpy().reshape(nw, 16)
            recs.append(st[:, :7].copy())
    r = np.stack(recs[len(layers):])            # drop the first (warm-up) rotation
    t0 = r[:, :, 0].min(axis=1, keepdims=True)[:, :, None]
    rel = (r - t0).astype(np.float64)            # cycles since the earliest wave start of the block
    out[name] = dict(mean_per_stamp=rel.mean(axis=(0, 1)).tolist(), max_per_stamp=rel.max(axis=1).mean(axis=0).tolist())
    print(name, "mean cycles since block start per stamp [start, issued, chunk0, chunks, shuffles, barrier, end]:")
    print("   mean over waves:", [int(v) for v in out[name]["mean_per_stamp"]])
    print("   slowest wave   :", [int(v) for v in out[name]["max_per_stamp"]])
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "timeline.json"), "w"), indent=1)
