"""Produce gemlite_amd/configs/mi355x.json: helper.autotune_layer (cold weights, device time) over the common LLM
shapes x the reference's M buckets.  Run on the MI355X; the result lands in gpurun_out/<tag>/mi355x.json and is then
committed under gemlite_amd/configs/ (the library autoloads it by device name, core.autoload_default_config)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemlite_amd
from gemlite_amd import GemLiteLinear, DType, core, helper

os.environ["GEMLITE_HIP_NO_DEFAULT_CONFIG"] = "1"
dev = torch.device("cuda:0")
out_dir = os.path.join("gpurun_out", os.environ.get("GL_TAG", "run"))
os.makedirs(out_dir, exist_ok=True)
SHAPES = [(4096, 4096), (8192, 8192), (14336, 4096), (4096, 14336), (4096, 11008), (11008, 4096), (5120, 5120), (13824, 5120),
          (5120, 13824), (1536, 8960), (8960, 1536), (28672, 8192), (8192, 28672), (16384, 16384)]  # (N, K)
MS = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024)
g = torch.Generator(device=dev).manual_seed(0)
report = []
core.GemLiteLinear.reset_config()
t0 = time.time()
for tdt, code in ((torch.float16, DType.FP16),):  # bf16 shares the table key with fp16 (core.py:141-145 aliases it)
    for nbits in (4,):
        for N, K in SHAPES:
            W_q = torch.randint(0, 2 ** nbits, (N, K), generator=g, dtype=torch.int32, device=dev).to(torch.uint8)
            s = (torch.rand(N * K // 128, 1, generator=g, device=dev) * 0.01 + 0.001).to(tdt)
            z = (torch.rand(N * K // 128, 1, generator=g, device=dev) * 15).to(tdt)
            lin = GemLiteLinear(nbits, 128, K, N, code, code).pack(W_q, s, z, None)
            del W_q
            ms = MS if N * K < 16384 * 16384 else (1, 16, 256)
            res = helper.autotune_layer(lin, batch_sizes=ms, iters=20, cold=True)
            for M, r in res.items():
                report.append(dict(dtype=str(tdt)[6:], nbits=nbits, N=N, K=K, M=M, **r))
                print(json.dumps(report[-1]), flush=True)
            del lin
            torch.cuda.empty_cache()
# keep only entries that beat the planner by > 5 % (the rest would just pin today's defaults)
table = {}
for fam, entries in core.GEMLITE_HIP_CONFIG_CACHE.items():
    for key, e in entries.items():
        table.setdefault(fam, {})[key] = e
keep = {}
for r in report:
    if r.get("default_us") and r["us"] < 0.95 * r["default_us"] and list(r["tuning"]) != [0, 0, 0, 0]:
        fam = core.config_family(-1, r["M"], r["nbits"])
        tid = (1 if r["dtype"] in ("float16", "bfloat16") else 0) * 100 + r["nbits"]
        key = core.config_key(r["M"], r["N"], r["K"], 128, 32 // r["nbits"], tid)
        if fam in table and key in table[fam]:
            keep.setdefault(fam, {})[key] = table[fam][key]
json.dump(keep, open(os.path.join(out_dir, "mi355x.json"), "w"), indent=0, sort_keys=True)
json.dump(report, open(os.path.join(out_dir, "autotune_report.json"), "w"), indent=0)
print(f"kept {sum(len(v) for v in keep.values())} of {len(report)} cells in {time.time() - t0:.0f} s")
