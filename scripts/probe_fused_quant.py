"""Round 4: `layer(x)` of the dynamically quantised A8W8 layers with the activation quantisation inside the matmul launch
(csrc/gl_coopquant.h: producer blocks + row flags, one launch; opt-in gemlite_amd.core.FUSE_ACT_QUANT_ROWS) against quantiser + matmul
(two launches, the default); graph-replayed time per layer(x), outputs compared bitwise.  Result (profiles/r04/probe_fused_quant_v*.log):
bit-identical, 3.3-7 us slower — the in-launch hand-off is ~5 dependent device-scope memory round trips.
    python scripts/probe_fused_quant.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gemlite_amd.core as core
from gemlite_amd import _hip

lib = _hip.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
W = bench.WORKLOADS
names = []
for m in (2, 8, 16, 32, 64):
    W[f"a8w8_4096_m{m}"] = (4096, 4096, 8, 4096, m, "int8", 32, "mfma" if m > 64 else "hbm")
    W[f"fp8_4096_m{m}"] = (4096, 4096, 8, 4096, m, "fp8w8", 32, "mfma" if m > 64 else "hbm")
    names += [f"a8w8_4096_m{m}", f"fp8_4096_m{m}"]
for m in (16,):
    W[f"a8w8_8192_m{m}"] = (8192, 8192, 8, 8192, m, "int8", 8, "mfma" if m > 64 else "hbm")
    names.append(f"a8w8_8192_m{m}")
W["fp8_16384_m16"] = (16384, 16384, 8, 16384, 16, "fp8w8", 2, "hbm")
names.append("fp8_16384_m16")
only = sys.argv[1:]
for name in names:
    if only and name not in only:
        continue
    ys = {}
    for fused in (False, True, False, True):
        core.FUSE_ACT_QUANT_ROWS = fused
        try:
            r = bench.Runner(name, dev, lib, e2e=True)
            y = r.call(r.mods[0])
            torch.cuda.synchronize()
            ys.setdefault(fused, y)
            c_us, n, el = r.chained_us_per_launch(min_seconds=0.15)
            print(json.dumps(dict(workload=name, fused=fused, kernel=r.kernel_name(), e2e_us=round(c_us, 3),
                                  equal=bool(torch.equal(ys[fused], ys.get(False, y))))), flush=True)
            del r
        except Exception as e:
            print(json.dumps(dict(workload=name, fused=fused, error=f"{type(e).__name__}: {e}"[:300])), flush=True)
        finally:
            core.FUSE_ACT_QUANT_ROWS = False
        torch.cuda.empty_cache()
