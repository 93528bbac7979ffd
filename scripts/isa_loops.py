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


def _find_objdump():
    """llvm-objdump of the ROCm toolchain: $ROCM_PATH, hipconfig --rocmpath, /opt/rocm, then PATH."""
    import shutil
    roots = [os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH")]
    try:
        roots.append(subprocess.run(["hipconfig", "--rocmpath"], capture_output=True, text=True, timeout=20).stdout.strip())
    except Exception:
        pass
    roots.append("/opt/rocm")
    for r in roots:
        if r and os.path.exists(os.path.join(r, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(r, "lib", "llvm", "bin", "llvm-objdump")
    return shutil.which("llvm-objdump")


OBJDUMP = _find_objdump()
# kernels whose main loops must never drain the queue (mangled-name fragments)
HOT = ("gemm_wn_mma_kernel", "gemm_a8w8_lds_kernel", "gemm_a8w8_mma_kernel", "gemm_mx_mma_kernel", "gemv_wn_kernel", "gemv_mfma_kernel",
       "gemv_w4_decode_kernel", "gemm_wn_direct_kernel", "a8w8_rows_kernel",
       # round 4 kernels (the list had not been extended) and round 5
       "gemv_w4_decode3_kernel", "a8w8_decode_kernel", "a16w8_decode_kernel", "a16w8_rows_kernel", "a8wn_rows_kernel", "gemm_a8w8_sq_kernel",
       "gemm_mx_sq_kernel", "gemm_mx_tile_kernel", "mx_rows_kernel", "nvfp4_rows_kernel", "gemv_a8wn_kernel", "gemm_w4_rows_kernel", "w8_rows_lds_kernel",
       "gemm_a8w8_sq128_kernel")
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


def natural_loops_with_drain(lines):
    """lines: [(address, text)] of one kernel.  Real loop detection (round 4): basic blocks, dominators, back edges (u -> v with v
    dominating u), natural loop bodies.  Every `s_waitcnt vmcnt(0)` is charged to the INNERMOST natural loop that contains it; a
    loop is reported when it also holds >= 4 MFMA / dot2 instructions.  (Round 3 took every backward branch for a loop: block
    placement puts shared exit blocks and the poll loop of the data-as-flag combine — which drains on purpose — behind branches
    that jump backwards across the K loop.)"""
    n = len(lines)
    if n == 0:
        return []
    index = {a: i for i, (a, _) in enumerate(lines)}
    br = {}  # instruction index -> (target index or None, falls_through)
    for i, (a, t) in enumerate(lines):
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\d+)", t)
        if m:
            offw = int(m.group(2))
            if offw >= 32768:
                offw -= 65536
            br[i] = (index.get(a + 4 + offw * 4), m.group(1) != "s_branch")
        elif t.startswith("s_endpgm") or t.startswith("s_setpc"):
            br[i] = (None, False)
    leaders = {0}
    for i, (tgt, _) in br.items():
        if tgt is not None:
            leaders.add(tgt)
        if i + 1 < n:
            leaders.add(i + 1)
    starts = sorted(leaders)
    bidx = {}
    for b, st in enumerate(starts):
        en = starts[b + 1] if b + 1 < len(starts) else n
        for i in range(st, en):
            bidx[i] = b
    nb = len(starts)
    succ = [[] for _ in range(nb)]
    for b, st in enumerate(starts):
        last = (starts[b + 1] if b + 1 < nb else n) - 1
        if last in br:
            tgt, ft = br[last]
            if tgt is not None:
                succ[b].append(bidx[tgt])
            if ft and last + 1 < n:
                succ[b].append(bidx[last + 1])
        elif last + 1 < n:
            succ[b].append(bidx[last + 1])
    pred = [[] for _ in range(nb)]
    for b in range(nb):
        for c in succ[b]:
            pred[c].append(b)
    # dominators as bitsets (iterative; blocks are in address order, close to reverse post-order)
    full = (1 << nb) - 1
    dom = [full] * nb
    dom[0] = 1
    changed = True
    while changed:
        changed = False
        for b in range(1, nb):
            d = full
            for q in pred[b]:
                d &= dom[q]
            d |= 1 << b
            if d != dom[b]:
                dom[b] = d
                changed = True
    loops = []  # (header, set of blocks)
    for u in range(nb):
        for v in succ[u]:
            if (dom[u] >> v) & 1:  # back edge u -> v
                body, stack = {v, u}, [u]
                while stack:
                    x = stack.pop()
                    if x == v:
                        continue
                    for q in pred[x]:
                        if q not in body:
                            body.add(q)
                            stack.append(q)
                loops.append((v, body))
    out = {}
    for i, (a, t) in enumerate(lines):
        if not re.match(r"s_waitcnt vmcnt\(0\)", t):
            continue
        b = bidx[i]
        inner = None
        for (hdr, body) in loops:
            if b in body and (inner is None or len(body) < len(inner[1])):
                inner = (hdr, body)
        if inner is not None:
            key = (inner[0], frozenset(inner[1]))
            out[key] = out.get(key, 0) + 1
    res = []
    for (hdr, body), drains in out.items():
        instr = [i for i in range(n) if bidx[i] in body]
        hot = sum(("v_mfma" in lines[i][1]) or lines[i][1].startswith("v_dot2") for i in instr)
        if hot >= 4:
            res.append((lines[starts[hdr]][0], len(instr), hot, drains))
    return res


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
            for (hdr_addr, n_instr, hot, drains) in natural_loops_with_drain(lines):
                (known if any(al in fn for al in ALLOW) else bad).append((fn, hex(hdr_addr), n_instr, hot, drains))
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
    if not OBJDUMP or not os.path.exists(LIB):
        print(f"isa_loops: cannot check (llvm-objdump: {OBJDUMP}, library present: {os.path.exists(LIB)})")
        sys.exit(2)
    seen, bad, known = loops_with_drain()
    if seen == 0:
        print("isa_loops: no gfx950 kernel of the hot families found in the library — nothing was checked")
        sys.exit(2)
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
