#!/usr/bin/env python3
"""ISA guard: no kernel of libgemlite_hip.so may use scratch memory or spill registers.

`make -C gemlite_amd/csrc` keeps hipcc's -Rpass-analysis=kernel-resource-usage remarks of every translation unit in
gemlite_amd/csrc/build/<unit>.remarks; this script parses them and fails (exit 1) on ScratchSize > 0, VGPR / SGPR spills or a
dynamic stack in any kernel (round 2 met three silent hipcc problems of this kind: 340 spilled accumulators in an int8 loop, a
scratch reload = s_waitcnt vmcnt(0) inside a DMA pipeline, 16 accumulators shuffled through scratch at a branch between two
epilogues).  Used by __graft_entry__.build() and tests/test_host_cpu.py.    python scripts/isa_guard.py [--list]
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Kernels known to spill when the guard was introduced (round 3) — fallback paths, none of them on a BASELINE configuration:
# the round-1 LDS-staged streaming kernel (groups of 32, manual GEMM_SPLITK at 33..64 rows) and the 4-row form of the A8Wn decode
# kernel (128 registers at 1024 threads).  Anything else with scratch or spills fails the build.
KNOWN_SPILLERS = (r"gemm_wn_stream_kernel<", r"gemv_a8wn_kernel<gl::\w+, [24], \d+, 4(, false)?>")
BUILD = os.path.join(ROOT, "gemlite_amd", "csrc", "build")


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"unit": os.path.basename(path)[:-8], "mangled": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            v = m.group(2)
            cur[m.group(1).strip()] = int(v) if v.isdigit() else v
    return rows


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:
        return names


def check():
    files = sorted(glob.glob(os.path.join(BUILD, "*.remarks")))
    if not files:
        raise SystemExit("no build/*.remarks: run `make -C gemlite_amd/csrc` first")
    rows = [r for f in files for r in parse(f)]
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = n
    # (SGPR spills go to VGPR lanes, not to memory: reported by --list, not an error)
    bad = [r for r in rows if r.get("ScratchSize", 0) or r.get("VGPRs Spill", 0) or str(r.get("Dynamic Stack", "False")) != "False"]
    known = [r for r in bad if any(re.search(k, r["name"]) for k in KNOWN_SPILLERS)]
    return rows, [r for r in bad if r not in known], known


if __name__ == "__main__":
    rows, bad, known = check()
    if "--list" in sys.argv:
        for r in rows:
            print(f"{r['unit']:22s} V{r.get('VGPRs', '?'):>4} A{r.get('AGPRs', '?'):>4} S{r.get('TotalSGPRs', '?'):>4} scratch {r.get('ScratchSize', '?'):>4}  {r['name'][:140]}")
    print(f"isa_guard: {len(rows)} kernels in {len(set(r['unit'] for r in rows))} units; scratch / spills: {len(bad)} new, {len(known)} known (fallback kernels)")
    for r in bad:
        print("  BAD", r["unit"], r["name"][:160], {k: r.get(k) for k in ("VGPRs", "ScratchSize", "VGPRs Spill", "Dynamic Stack")})
    sys.exit(1 if bad else 0)
