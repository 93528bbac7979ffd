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
