"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs (one directory per pass)."""
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])
            per[key] += float(r["Counter_Value"])
            meta[(r["Dispatch_Id"], r["Kernel_Name"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Grid_Size"], r["Workgroup_Size"])
        for (did, kn, cn), v in per.items():
            tot[kn][cn].append(v)
        for (did, kn), m in meta.items():
            tot[kn]["_dur_ns"].append(m[0])
            tot[kn]["_vgpr"].append(float(m[1]))
for kn, cs in tot.items():
    if "gemlite" in kn or "gl::" in kn:
        print(kn[:110])
        for cn in sorted(cs):
            v = cs[cn]
            print(f"    {cn:32s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
        # derived (MI355X: 32 shader engines, 1024 SIMDs; SQ_BUSY_CYCLES is summed over the SEs, SQ_WAVE_CYCLES and the
        # SQ_WAIT_* / SQ_ACTIVE_INST_* family count quad-cycles): the clock the kernel actually ran at, and how busy the
        # matrix pipes were AT THAT CLOCK
        mean = lambda k: sum(cs[k]) / len(cs[k]) if k in cs and cs[k] else None
        busy, dur, mf, wc = mean("SQ_BUSY_CYCLES"), mean("_dur_ns"), mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("SQ_WAVE_CYCLES")
        if busy and dur:
            cyc = busy / 32.0
            line = f"    => effective clock {cyc / dur:.2f} GHz ({cyc:.0f} cycles in {dur / 1e3:.1f} us)"
            if mf:
                line += f"; MFMA pipes busy {100.0 * mf / (cyc * 1024):.1f} % of the SIMD cycles"
            for k, nm in (("SQ_WAIT_ANY", "parked (waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_ACTIVE_INST_ANY", "issuing")):
                if wc and mean(k) is not None:
                    line += f"; {nm} {100.0 * mean(k) / wc:.0f} %"
            print(line)
