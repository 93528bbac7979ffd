"""Round 6 (code health: "each round adds a rule; nothing re-validates the old ones"): one report over the planner sweeps of this round —
for every (bit width, M, N, K) cell: the default kernel's time, the best measured candidate, the regret.   python scripts/r6/planner_regret.py <dir with the probe logs> > report.json"""
import glob, json, os, sys

d = sys.argv[1]
cells = []
for f in sorted(glob.glob(os.path.join(d, "probe_m1_shapes_*.log"))):   # scripts/probe_m1_shapes.py: default + candidates
    bits = 2 if "_w2" in f else 4
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l)
            cells.append(dict(src=os.path.basename(f), bits=bits, M=r["M"], N=r["N"], K=r["K"], default_us=r["default"][0], default_kernel=r["default"][1],
                              best_us=r["best"][1], best=r["best"][0] + " " + r["best"][2]))
for f in sorted(glob.glob(os.path.join(d, "probe_rows5_*.log"))):       # scripts/probe_rows5.py: r4 / rows5 / mma / default
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l)
            us = {k: v for k, v in r["us"].items() if isinstance(v, (int, float)) and k != "default"}
            if not us or r["us"].get("default") is None:
                continue
            b = min(us, key=us.get)
            cells.append(dict(src=os.path.basename(f), bits=4, gs=r["gs"], M=r["M"], N=r["N"], K=r["K"], default_us=r["us"]["default"], default_kernel=r["default_kernel"], best_us=us[b], best=b))
for f in sorted(glob.glob(os.path.join(d, "probe_mma_narrow_shapes*.log"))):   # scripts/probe_mma_narrow_shapes.py: auto vs forced tile candidates
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l)
            us = {k: v for k, v in r["us"].items() if v and k != "auto"}
            if not us or not r["us"].get("auto"):
                continue
            b = min(us, key=us.get)
            cells.append(dict(src=os.path.basename(f), bits=4, M=r["M"], N=r["N"], K=r["K"], default_us=r["us"]["auto"], default_kernel=r["auto_kernel"], best_us=us[b], best=b))
for c in cells:
    c["regret"] = round(c["default_us"] / c["best_us"], 3)
worst = sorted(cells, key=lambda c: -c["regret"])[:15]
n = len(cells)
summary = dict(cells=n, within_3pct=sum(c["regret"] <= 1.03 for c in cells), within_6pct=sum(c["regret"] <= 1.06 for c in cells), within_10pct=sum(c["regret"] <= 1.10 for c in cells),
               mean_regret=round(sum(c["regret"] for c in cells) / max(1, n), 4), worst=worst)
print(json.dumps(dict(summary=summary, cells=cells), indent=1))
