#!/usr/bin/env python3
"""Summarise the counter passes of scripts/pmc_official.sh: per workload, the dominant library kernel's HBM-side bytes per launch
(FETCH_SIZE x correction + WRITE_SIZE, KiB -> bytes; the correction factors come from the calibration kernels of
scripts/ubench/fetch_calib.hip measured in the same run) and its matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x
1024 SIMDs) at the clock the kernel actually ran at: SQ_BUSY_CYCLES / 32 shader engines / duration).  Writes
<outdir>/pmc_traffic.json and <outdir>/mfma_util.json (the builder copies them to profiles/).   usage: pmc_official.py <outdir>"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def load(d):
    """{kernel: {counter: [value per dispatch]}} + durations"""
    tot = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per, meta = collections.defaultdict(float), {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
            meta[(r["Dispatch_Id"], r["Kernel_Name"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for (did, kn, cn), v in per.items():
            tot[kn][cn].append(v)
        for (did, kn), dur in meta.items():
            tot[kn]["_dur_ns"].append(dur)
    return tot


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


# ---- calibration: counter (KiB) per KiB actually moved
calib = {}
ca, cb = load(os.path.join(out, "calib_A")), load(os.path.join(out, "calib_B"))
moved_kib = 512 * 1024
for kn, cs in ca.items():
    if "k_read" in kn and "FETCH_SIZE" in cs:
        calib[kn.split("(")[0]] = {"FETCH_SIZE_per_KiB_read": round(med(cs["FETCH_SIZE"]) / moved_kib, 4)}
for kn, cs in cb.items():
    if "k_write" in kn and "WRITE_SIZE" in cs:
        calib[kn.split("(")[0]] = {"WRITE_SIZE_per_KiB_written": round(med(cs["WRITE_SIZE"]) / moved_kib, 4)}
f16 = calib.get("k_read16", {}).get("FETCH_SIZE_per_KiB_read")
f4 = calib.get("k_read4_rows", {}).get("FETCH_SIZE_per_KiB_read")
wsc1 = calib.get("k_write16_sc1", {}).get("WRITE_SIZE_per_KiB_written")
w16 = calib.get("k_write16", {}).get("WRITE_SIZE_per_KiB_written")
traffic = {"_calibration": calib,
           "_note": "bytes per launch = FETCH_SIZE KiB / (FETCH_SIZE per KiB of a 16 B/lane streaming read, measured) x 1024 + WRITE_SIZE KiB / "
                    "(WRITE_SIZE per KiB of plain 16-byte stores, measured) x 1024; median over the launches of the workload's dominant "
                    "library kernel, eager launches over rotating layers.  k_read4_rows = the tile kernel's B-operand pattern."}
# stamp: the PLANNER's kernel name of each workload (from the bench line the pass printed) + the commit the passes ran on (GL_COMMIT, set by
# the caller: the GPU box has no .git).  bench.py prints `traffic: null` when the planner's current choice differs from the stamp.
commit = os.environ.get("GL_COMMIT", "unknown")


def planner_kernel(w):
    try:
        for ln in open(os.path.join(out, w + "_A.log")):
            if ln.startswith("{") and '"roofline"' in ln:
                return json.loads(ln)["roofline"].get("kernel")
    except Exception:
        pass
    return None


traffic["_stamp"] = {}
util = {"_stamp": {}, "_note": "matrix-pipe busy share of the SIMD cycles at the clock the kernel ran at (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024)), "
                 "effective clock in GHz, VALU instructions per MFMA; eager launches under rocprofv3 --pmc"}
for da in sorted(glob.glob(os.path.join(out, "*_A"))):
    w = os.path.basename(da)[:-2]
    if w == "calib":
        continue
    A, B = load(da), load(os.path.join(out, w + "_B"))
    # dominant library kernel = the gl:: kernel with the largest total duration
    ks = [(sum(cs["_dur_ns"]), kn) for kn, cs in A.items() if "gl::" in kn and "act_quant" not in kn and "pack_over_cols" not in kn and "noop" not in kn]
    if not ks:
        continue
    kn = max(ks)[1]
    cs = A[kn]
    fetch_kib = med(cs.get("FETCH_SIZE", []))
    write_kib = med(B.get(kn, {}).get("WRITE_SIZE", []))
    rec = {"kernel": kn[:100], "launches": len(cs["_dur_ns"]), "dur_us_under_pmc": round(med(cs["_dur_ns"]) / 1e3, 2), "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib}
    if fetch_kib is not None and f16:
        b = fetch_kib / f16 * 1024
        if write_kib is not None and w16:
            b += write_kib / w16 * 1024
        traffic[w] = int(b)
    traffic.setdefault("_detail", {})[w] = rec
    traffic["_stamp"][w] = util["_stamp"][w] = {"kernel": planner_kernel(w), "commit": commit}
    busy, mf, dur = med(cs.get("SQ_BUSY_CYCLES", [])), med(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), med(cs["_dur_ns"])
    if busy and dur:
        cyc = busy / 32.0
        u = {"effective_clock_ghz": round(cyc / dur, 3)}
        if mf:
            u["mfma_busy"] = round(mf / (cyc * 1024), 4)
        iv, im, wc = med(cs.get("SQ_INSTS_VALU", [])), med(cs.get("SQ_INSTS_MFMA", [])), med(cs.get("SQ_WAVE_CYCLES", []))
        if iv and im:
            u["valu_per_mfma"] = round(iv / im, 2)
        for k, nm in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "issue_stalled"), ("SQ_ACTIVE_INST_ANY", "issuing")):
            if wc and med(cs.get(k, [])) is not None:
                u[nm] = round(med(cs[k]) / wc, 3)
        util[w] = u
# per-kernel means of every counter, one small CSV per pass (the raw per-dispatch files are deleted by pmc_official.sh afterwards)
for d in sorted(glob.glob(os.path.join(out, "*_[AB]"))):
    if not os.path.isdir(d):
        continue
    t = load(d)
    with open(d + "_counters_by_kernel.csv", "w") as f:
        f.write("Kernel_Name,Counter_Name,samples,mean_value\n")
        for kn in sorted(t):
            for cn in sorted(t[kn]):
                v = t[kn][cn]
                f.write('"%s",%s,%d,%.3f\n' % (kn.replace('"', "'"), cn, len(v), sum(v) / max(1, len(v))))
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
json.dump(util, open(os.path.join(out, "mfma_util.json"), "w"), indent=1)
print(json.dumps({"calibration": calib, "traffic": {k: v for k, v in traffic.items() if not k.startswith("_")}, "mfma_util": {k: v for k, v in util.items() if not k.startswith("_")}}, indent=1))
