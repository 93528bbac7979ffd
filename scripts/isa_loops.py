#!/usr/bin/env python3
"""ISA guard, part 2: no `s_waitcnt vmcnt(0)` inside the K loops of the hot kernels.

Extracts every gfx950 code object from libgemlite_hip.so (the clang offload bundles are uncompressed: header + entries), disassembles
it with llvm-objdump and, per kernel of the hot families, looks at every LOOP (a backward branch and its target) that contains matrix
or dot-product instructions: a full drain (`s_waitcnt vmcnt(0)`) inside such a loop means a register request is waited for right
behind the DMA / prefetch queue — the failure that cost round 2 a 4x slowdown once (a scratch reload) and a step-long stall once (a
tracked load answered with vmcnt(0)).    python scripts/isa_loops.py [--list]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gemlite_amd", "csrc", "libgemlite_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernels whose main loops must never drain the queue (mangled-name fragments)
HOT = ("gemm_wn_mma_kernel", "gemm_a8w8_lds_kernel", "gemm_a8w8_mma_kernel", "gemm_mx_mma_kernel", "gemv_wn_kernel", "gemv_mfma_kernel",
       "gemv_w4_decode_kernel", "gemm_wn_direct_kernel", "a8w8_rows_kernel")
# MFMA tile kernels with a known drain (baseline file, see its header): reported, not fatal; anything outside it fails the build.
KNOWN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa_loops_known.txt")
# Known offenders when the check was introduced (round 3): reported, not fatal.  The two-buffer loops of the decode kernels still
# get `s_waitcnt vmcnt(6)` followed a few instructions later by `vmcnt(0)` at the loop head from hipcc 7.2 (after unconditional
# priming, sched_barrier between the phases and compiling the timeline stores out — the remaining trigger was not found), so the
# second buffer's requests overlap only half of the arithmetic.  Anything else — every MFMA tile kernel — must be clean.
ALLOW = ("gemv_wn_kernel", "gemv_mfma_kernel", "gemv_w4_decode_kernel", "gemm_wn_direct_kernel")


def code_objects(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    for m in re.finditer(re.escape(magic), data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        off = p + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s > 0:
                yield data[p + o:p + o + s]


def loops_with_drain():
    bad, known, seen = [], [], 0
    for co in code_objects(LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        fn, lines = None, []

        def flush():
            nonlocal seen
            if not fn or not any(h in fn for h in HOT):
                return
            seen += 1
            addr = {}
            for i, (a, t) in enumerate(lines):
                addr[a] = i
            for i, (a, t) in enumerate(lines):
                m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", t)
                if not m:
                    continue
                # llvm-objdump prints the branch target as an absolute address comment: "// 000000001234: ..." is not available with
                # --no-show-raw-insn; the operand is a signed word offset relative to the next instruction
                offw = int(m.group(1) or m.group(2))
                if offw >= 32768:
                    offw -= 65536
                tgt = a + 4 + offw * 4
                if tgt <= a and tgt in addr:
                    body = [t2 for (_, t2) in lines[addr[tgt]:i + 1]]
                    hot = sum(("v_mfma" in t2) or t2.startswith("v_dot2") for t2 in body)  # arithmetic loops only (gather loops drain on purpose)
                    drains = sum(bool(re.match(r"s_waitcnt vmcnt\(0\)", t2)) for t2 in body)
                    if hot >= 4 and drains:
                        (known if any(al in fn for al in ALLOW) else bad).append((fn, hex(tgt), len(body), hot, drains))
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
    return seen, bad, known


if __name__ == "__main__":
    seen, bad, known = loops_with_drain()
    base = set(l.strip() for l in open(KNOWN_FILE) if l.strip() and not l.startswith("#")) if os.path.exists(KNOWN_FILE) else set()
    bad_fns = sorted(set(b[0] for b in bad))
    dem = dict(zip(bad_fns, subprocess.run(["c++filt"], input="\n".join(bad_fns), capture_output=True, text=True).stdout.strip().split("\n"))) if bad_fns else {}
    new = [b for b in bad if dem[b[0]] not in base]
    listed = sorted(set(dem[b[0]] for b in bad if dem[b[0]] in base))
    print(f"isa_loops: {seen} hot kernels checked; full vmcnt drain inside an MFMA / dot-product loop: {len(set(b[0] for b in new))} new kernels, "
          f"{len(listed)} MFMA-tile kernels on the known list (scripts/isa_loops_known.txt), {len(set(k[0] for k in known))} decode-family kernels known")
    for (fn, tgt, n, hot, d) in new[:40]:
        print(f"  DRAIN {dem[fn][:130]} loop@{tgt} ({n} instr, {hot} mfma / dot2, {d} x vmcnt(0))")
    gone = sorted(base - set(dem.values()))
    if gone:
        print(f"  ({len(gone)} kernels of the known list are clean now — remove them from scripts/isa_loops_known.txt)")
    sys.exit(1 if new else 0)
