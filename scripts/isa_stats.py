#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel of the built library (development aid; uses the code-object extraction of
isa_loops.py).  For every loop that contains MFMA / dot2 instructions: how many MFMA, VALU (by opcode), SALU, LDS, vector-memory and
wait instructions one trip issues — the static counterpart of the PMC figures `valu_per_mfma` / `issue_stalled`.
    python scripts/isa_stats.py 'gemm_wn_mma_kernelINS_8bf16_tagELi4ELi4ELi128'          (a fragment of the mangled name)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_loops as IL


def classify(t):
    op = t.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("ds_", )):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    frag = sys.argv[1]
    for co in IL.code_objects(IL.LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            asm = subprocess.run([IL.OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        fn, lines, out = None, [], []

        def flush():
            if not fn or frag not in fn:
                return
            addr = {a: i for i, (a, t) in enumerate(lines)}
            print(f"== {subprocess.run(['c++filt', fn], capture_output=True, text=True).stdout.strip()[:150]}  ({len(lines)} instructions)")
            for i, (a, t) in enumerate(lines):
                m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", t)
                if not m:
                    continue
                offw = int(m.group(1) or m.group(2))
                if offw >= 32768:
                    offw -= 65536
                tgt = a + 4 + offw * 4
                if tgt <= a and tgt in addr:
                    body = [t2 for (_, t2) in lines[addr[tgt]:i + 1]]
                    cls = collections.Counter(classify(t2) for t2 in body)
                    if cls["mfma"] + sum(t2.startswith("v_dot2") for t2 in body) < 4:
                        continue
                    ops = collections.Counter(t2.split()[0] for t2 in body if classify(t2) == "valu")
                    waits = [t2 for t2 in body if t2.startswith("s_waitcnt")]
                    print(f"  loop @{tgt:#x}: {len(body)} instr | " + " ".join(f"{k}={v}" for k, v in sorted(cls.items())) +
                          (f" | valu/mfma={cls['valu'] / cls['mfma']:.2f} salu/mfma={cls['salu'] / cls['mfma']:.2f} lds/mfma={cls['lds'] / cls['mfma']:.2f}" if cls["mfma"] else ""))
                    print("    valu: " + ", ".join(f"{k} x{v}" for k, v in ops.most_common(14)))
                    print("    waits: " + ", ".join(f"{k} x{v}" for k, v in collections.Counter(waits).most_common(8)))
        for line in asm.split("\n"):
            m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
            if m:
                flush()
                fn, lines = m.group(2), []
                continue
            m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if m and fn:
                lines.append((int(m.group(2), 16), m.group(1).strip()))
        flush()


if __name__ == "__main__":
    main()
