"""Regenerate the "Round-N summary" section of profiles/README.md from an official bench line + the rocprofv3 kernel stats of the same
command.   python scripts/make_profiles_summary.py profiles/r04/official [--boxes "1975 / 1957 ... GB/s (4.52 / ... µs)"]"""
import csv, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
off = sys.argv[1]
boxes = sys.argv[sys.argv.index("--boxes") + 1] if "--boxes" in sys.argv else None
d = json.load(open(os.path.join(root, off, "bench_default.json")))
r = d["roofline"]
# round 5: the printed line is compact; `bench.py --full-out bench_full.json` keeps the verbose blocks next to it
full = {}
if os.path.exists(os.path.join(root, off, "bench_full.json")):
    full = json.load(open(os.path.join(root, off, "bench_full.json"))).get("blocks", {})
avg = calls = mn = None
for row in csv.DictReader(open(os.path.join(root, off, "rocprof_bench_kernel_stats.csv"))):
    if "gemv_w4_decode3_kernel<gl::half_tag, true>" in row["Name"]:
        avg, calls, mn = float(row["AverageNs"]) / 1000, int(row["Calls"]), float(row["MinNs"]) / 1000
cb = d["cpu_baseline"]

head = (f"**value {d['value']:.0f} GB/s** (4096² M = 1 fp16, 64 cold layers = 572 MB per step, {d['ms_per_step'] * 1000:.1f} µs per step); `roofline`: "
        f"`{r['kernel']}` **{r['kernel_us']:.3f} µs** per launch (the timed region itself) = **{r['frac']:.4f}** of 8 TB/s; an empty kernel in the same graph "
        f"{r['empty_launch_us']:.2f} µs → the kernel's own part {r['kernel_us_minus_empty_launch']:.2f} µs; `rotation_ab`: 32 layers "
        f"{(r.get('rotation_ab') or d.get('rotation_ab'))['layers32_286MB_us']:.2f} µs vs 64 layers {(r.get('rotation_ab') or d.get('rotation_ab'))['layers64_572MB_us']:.2f} µs; sustained 6 s: "
        f"{(r.get('sustained') or d.get('sustained'))['us_per_launch']:.2f} µs; eager host cost {(r.get('eager') or d.get('eager'))['host_us_per_call']:.2f} µs per `layer(x)`; CPU baseline (port, {cb['cores']} threads) "
        f"{cb['value']:.3f} GB/s.  rocprofv3 (`rocprof_bench_kernel_stats.csv`): `gemv_w4_decode3_kernel<half,true>` avg {avg:.2f} µs over {calls} calls "
        f"(min {mn:.2f}; the average includes the eager passes and the 32-layer rotation block)." + (f"  The same command on the other boxes of the session: {boxes}." if boxes else ""))
rows = []


def add(block, key, b):
    tr, mu = b.get("traffic"), b.get("mfma_util")
    rate = f"{b['achieved']:.0f} {b.get('unit', 'GB/s')}" if "achieved" in b else "—"
    if not isinstance(mu, dict):
        mu = None
    rows.append(f"| `{block}`{(' / ' + key) if key else ''} | `{b['kernel']}` | {b['kernel_us']:.2f} | {rate} | {b['frac']:.3f} | "
                f"{('%.1f MB' % (tr / 1e6)) if tr else '—'} | "
                f"{('%.2f @ %.2f GHz, %.1f VALU/MFMA' % (mu['mfma_busy'], mu['effective_clock_ghz'], mu['valu_per_mfma'])) if mu else '—'} |")


add("roofline", "", r)
# round 5: the blocks live INSIDE `roofline` ({group: {label: compact block}}); rounds 1-4 had them as top-level `roofline_<group>` keys
for grp in ("m256", "m1_bf16", "fewrows", "cfg4", "cfg5", "prefill_m2048", "trend_m1", "mx_fewrows", "mx_m256"):
    blocks = full.get(grp) or (r.get(grp) if isinstance(r.get(grp), dict) else d.get("roofline_" + grp, {}))
    if isinstance(blocks, dict) and "kernel" in blocks:  # (old layout: a single block)
        blocks = {"": blocks}
    for k, b in (blocks or {}).items():
        if isinstance(b, dict) and "kernel" in b:
            add("roofline." + grp, k, b)
table = ("| block | kernel | µs per launch | algorithmic rate | fraction of peak | HBM traffic (PMC) | MFMA busy @ clock |\n|---|---|---|---|---|---|---|\n" + "\n".join(rows))
p = os.path.join(root, "profiles", "README.md")
s = open(p).read()
title = re.search(r"### Round-\d+ summary \(`[^`]*`\)", s).group(0)
i = s.index(title)
m = re.search(r"\n##+ ", s[i + 10:])
end = i + 10 + m.start() if m else len(s)
open(p, "w").write(s[:i] + title + "\n\n" + head + "\n\n" + table + "\n" + s[end:])
print(head[:200])
