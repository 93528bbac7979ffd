"""GPU debug helper: run a few small layers through chosen kernel variants and dump inputs/outputs (npz)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gemlite_amd
from gemlite_amd import GemLiteLinear, DType, _hip
from gemlite_amd.core import _hip_matmul
from oracle import gemlite_oracle as O

dev = "cuda:0"
out = {}
summary = []

def run(tag, N, K, nbits, gs, tdt, M, mt, tuning, special=None, w_mode_kind="fma"):
    W_q, s, z = O.gen_data(N, K, nbits, gs, seed=1)
    if special == "const":
        W_q[:] = 5; s[:] = 1.0; z[:] = 0.0
    lin = GemLiteLinear(nbits, gs, K, N, gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt], gemlite_amd.dtypes.TORCH_TO_DTYPE[tdt])
    zz = torch.from_numpy(z.astype(np.float32)).to(tdt).to(dev)
    lin.pack(torch.from_numpy(W_q).to(dev), torch.from_numpy(s.astype(np.float32)).to(tdt).to(dev),
             None if w_mode_kind == "sym" else zz, None, fma_mode=(w_mode_kind != "nofma"))
    if special == "unit":
        x = torch.zeros(M, K, dtype=tdt, device=dev)
        for m in range(M):
            x[m, (37 * (m + 1)) % K] = 1.0
    elif special == "ones" or special == "const":
        x = torch.ones(M, K, dtype=tdt, device=dev)
    else:
        x = torch.from_numpy(O.gen_x(M, K, seed=3).astype(np.float32)).to(tdt).to(dev)
    y = _hip_matmul(x, lin.W_q, lin.scales, lin.zeros, None, lin.get_meta_args(), mt, tuning)
    torch.cuda.synchronize()
    from tests.test_gpu_parity import _oracle_from_layer
    y_or = _oracle_from_layer(lin, x)
    yy = y.float().cpu().numpy().astype(np.float64)
    err = np.abs(yy - y_or)
    rel = err.mean() / max(np.abs(y_or).mean(), 1e-12)
    from gemlite_amd.core import _static_args
    a = _static_args(lin.W_q, lin.scales, lin.zeros, lin.get_meta_args())
    kname = _hip.load().gemlite_hip_kernel_name(_hip.C.byref(a)).decode()
    percol = err.mean(0)
    summary.append(dict(tag=tag, kernel=kname, rel=float(rel), max=float(err.max()),
                        err_by_col_mod4=[float(percol[i::4].mean()) for i in range(4)],
                        err_by_col_mod16=[float(percol[i::16].mean()) for i in range(16)],
                        bad_cols=int((percol > 1e-2 * np.abs(y_or).mean()).sum()), N=N))
    out[tag + "/y"] = yy; out[tag + "/y_or"] = y_or
    print(tag, kname, "rel=%.3g" % rel, flush=True)

H, B = torch.float16, torch.bfloat16
for tdt, tn in ((H, "f16"), (B, "bf16")):
    for special in (None, "const", "unit"):
        sp = special or "rand"
        run(f"narrow/{tn}/{sp}/M1", 256, 1024, 4, 128, tdt, 1, 1, (2, 0, 0, 0), special)
        run(f"wide/{tn}/{sp}/M1", 256, 1024, 4, 128, tdt, 1, 1, (4, 1, 0, 0), special)
        run(f"wide-split4/{tn}/{sp}/M1", 256, 1024, 4, 128, tdt, 1, 1, (4, 4, 0, 0), special)
        run(f"stream/{tn}/{sp}/M16", 256, 1024, 4, 128, tdt, 16, 3, (0, 1, 0, 0), special)
    run(f"narrow/{tn}/rand/M4", 256, 1024, 4, 128, tdt, 4, 1, (2, 0, 0, 0))
    run(f"narrow/{tn}/sym/M1", 256, 1024, 4, 128, tdt, 1, 1, (2, 0, 0, 0), None, "sym")
    run(f"narrow/{tn}/nofma/M1", 256, 1024, 4, 128, tdt, 1, 1, (2, 0, 0, 0), None, "nofma")
    run(f"narrow/{tn}/w2/M1", 256, 1024, 2, 128, tdt, 1, 1, (2, 0, 0, 0))
    run(f"cfgA-narrow/{tn}/M1", 4096, 4096, 4, 128, tdt, 1, 1, (2, 0, 0, 0))
    run(f"cfgA-wide/{tn}/M1", 4096, 4096, 4, 128, tdt, 1, 1, (4, 0, 0, 0))
    run(f"cfgA-stream/{tn}/M64", 4096, 4096, 4, 128, tdt, 64, 3, (0, 0, 0, 0))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "debug_dump.npz"), **{k: v for k, v in out.items() if v.size <= 4096 * 16})
json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "debug_summary.json"), "w"), indent=1)
