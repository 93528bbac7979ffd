#!/usr/bin/env python3
"""ISA guard, part 3 (round 4): registers loaded by inline-asm buffer loads must not be touched before the counted wait that covers them.

The MFMA tile kernel (gemm_wn_mma_kernel.inc) issues its weight / metadata loads from inline asm and retires them by hand with counted
`s_waitcnt vmcnt(N)`; hipcc does not know these registers are in flight and may copy, spill or reuse one between the load and the
wait (cdna_hip_programming.md §5.7 item 1).  This script replays the wave's memory queue over the generated code of every
gemm_wn_mma_kernel instantiation — function start to the end of the K loop, then the K loop a second time (the ring wraps around the
back edge) — and reports any instruction that reads or writes the destination of a load that is still outstanding at that point:
loads return in order, so at `s_waitcnt vmcnt(k)` everything but the newest k requests has landed.
    python scripts/isa_asmloads.py            exit 0: clean | 1: a destination is touched early | 2: nothing could be checked
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_loops as L

FAMILIES = ("gemm_wn_mma_kernel", "gemm_w4_rows_kernel", "gemv_wn_kernel", "w8_rows_lds_kernel")
# gemv_wn_kernel (round 6: gvw::ring2_run): the prologue requests up to two chunks behind uniform branches and the last one to three chunks run in
# straight-line code BEHIND the loop — replayed in address order (no branch is followed), prologue and tail included, every conditional request
# issued and every run of alternative waits taken at its weakest member
LINEAR_WITH_TAIL = ("gemv_wn_kernel",)
ALTERNATIVE_WAITS = ("gemm_w4_rows_kernel", "gemv_wn_kernel", "w8_rows_lds_kernel")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def first_operand_regs(text):
    ops = text.split(None, 1)[1] if " " in text else ""
    return regs_of(ops.split(",")[0])


def replay(lines, seq, weakest_of_alternatives=False):
    """seq: instruction indices in execution order.  Returns [(address, text, registers)] of early touches.
    weakest_of_alternatives (gemm_w4_rows_kernel, round 5): that kernel picks ONE of up to four counted waits with uniform branches (what sits
    behind the awaited piece depends on how many chunks are left); a linear replay would execute all four, vmcnt(0) included.  A run of
    `s_waitcnt vmcnt` separated only by scalar branch / compare instructions is therefore replayed as its WEAKEST member — the steady-state
    path, on which every conditional request is issued — which can only raise alarms, never hide one."""
    queue, bad = [], []  # queue: oldest first; each entry = set of destination registers (empty: store / LDS-DMA)
    skip_until = -1
    for pos, i in enumerate(seq):
        if pos <= skip_until:
            continue
        a, t = lines[i]
        op = t.split()[0]
        m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
        if m:
            k = int(m.group(1))
            if weakest_of_alternatives:
                nxt = pos + 1
                while nxt < len(seq) and nxt - pos < 24:
                    t2 = lines[seq[nxt]][1]
                    m2 = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t2)
                    if m2:
                        k = max(k, int(m2.group(1)))
                        skip_until = nxt
                    elif not t2.startswith(("s_cbranch", "s_branch", "s_cmp", "s_and", "s_or", "s_cselect", "s_mov", "s_nop", "s_xor")):
                        break
                    nxt += 1
            if len(queue) > k:
                queue = queue[len(queue) - k:] if k else []
            continue
        vmem = op.startswith(("buffer_", "global_", "flat_", "scratch_"))
        touched = regs_of(t)
        if vmem and "load" in op and " lds" not in t:
            dest = first_operand_regs(t)
            srcs = touched - dest
        else:
            dest, srcs = set(), touched
        inflight = set().union(*queue) if queue else set()
        clash = (srcs | dest) & inflight if vmem and "load" in op else touched & inflight
        if clash and op == "v_mad_u64_u32":
            # hipcc uses v_mad_u64_u32 for 32-bit multiply-adds and leaves the HIGH half of the 64-bit addend undefined — any register,
            # in-flight ones included.  That half only reaches the high half of the result: not a read of the loaded value if the high
            # result register is overwritten before anything reads it.
            m3 = re.match(r"v_mad_u64_u32 v\[(\d+):(\d+)\], [^,]+, [^,]+, [^,]+, v\[(\d+):(\d+)\]", t)
            if m3 and clash == {int(m3.group(4))}:
                hi = int(m3.group(2))
                dead = True  # (not read within the next 200 instructions counts as dead)
                for nxt in seq[pos + 1:pos + 200]:
                    t2 = lines[nxt][1]
                    ops2 = t2.split(None, 1)[1] if " " in t2 else ""
                    parts = ops2.split(",")
                    if hi in regs_of(",".join(parts[1:])) or (t2.split()[0].startswith(("buffer_store", "global_store", "ds_write", "scratch_store")) and hi in regs_of(ops2)):
                        dead = False  # read first
                        break
                    if hi in regs_of(parts[0]):
                        break  # overwritten first
                if dead:
                    clash = set()
        if clash and op.startswith("v_pk_") and "op_sel_hi:[1,0]" in t and "op_sel:" not in t:
            # packed fp32 with the SECOND source broadcast from its low register (op_sel_hi[1] = 0, op_sel[1] = 0): the high register of that
            # 64-bit operand is encoded but never read
            m4 = re.match(r"v_pk_\w+ v\[\d+:\d+\], [^,]+, v\[(\d+):(\d+)\]", t)
            if m4 and clash == {int(m4.group(2))}:
                clash = set()
        if clash:
            bad.append((a, t, sorted(clash)))
        if vmem and "atomic" not in op or (vmem and "atomic" in op):
            queue.append(dest)
    return bad


def check():
    seen, reports = 0, []
    for co in L.code_objects(L.LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            asm = subprocess.run([L.OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        fn, lines = None, []

        def flush():
            nonlocal seen
            if not fn or not any(h in fn for h in FAMILIES) or not lines:
                return
            seen += 1
            # the K loop = the natural loop with the most MFMAs; approximated by the last backward branch that spans >= 16 MFMAs
            # and whose body holds no s_endpgm: indices [lo, hi]
            index = {a: i for i, (a, _) in enumerate(lines)}
            best = None
            for i, (a, t) in enumerate(lines):
                m = re.match(r"(?:s_cbranch_\w+|s_branch)\s+(\d+)", t)
                if not m:
                    continue
                offw = int(m.group(1))
                if offw >= 32768:
                    offw -= 65536
                tgt = index.get(a + 4 + offw * 4)
                if tgt is None or tgt > i:
                    continue
                body = lines[tgt:i + 1]
                if any(x[1].startswith("s_endpgm") for x in body):
                    continue
                hot = sum(("v_mfma" in x[1] or "v_dot2" in x[1]) for x in body)
                if hot >= 16 and (best is None or hot > best[2] or (hot == best[2] and tgt == best[0])):  # (same head, later back edge: the longer body — a loop whose request block sits behind a conditional back edge)
                    best = (tgt, i, hot)
            if best is None:
                return
            lo, hi, _ = best
            # first pass: straight-line from the top, following UNCONDITIONAL forward branches (hipcc rotates some loops: the preheader ends in an
            # s_branch into the middle of the body — the ring form of the rows kernel with four pieces per chunk); then full iterations
            linear_tail = any(k in fn for k in LINEAR_WITH_TAIL)
            seq, i = [], 0
            while i <= hi:
                seq.append(i)
                m = re.match(r"s_branch\s+(\d+)", lines[i][1])
                if m and not linear_tail:
                    offw = int(m.group(1))
                    tgt = index.get(lines[i][0] + 4 + offw * 4) if offw < 32768 else None
                    if tgt is not None and i < tgt <= hi:
                        i = tgt
                        continue
                # round 6 (w8_rows_lds_kernel): hipcc moves a conditional prologue request ("if (npieces > 1) issue_x(1, 1)") OUT of line — a
                # conditional forward branch to a block behind the loop that issues the requests and branches back to the instruction after
                # its origin.  The steady-state path takes it: splice the block in where it executes.
                m = re.match(r"s_cbranch_\w+\s+(\d+)", lines[i][1])
                if m and not linear_tail and int(m.group(1)) < 32768:
                    tgt = index.get(lines[i][0] + 4 + int(m.group(1)) * 4)
                    if tgt is not None and tgt > hi:
                        blk, e = [], tgt
                        while e < len(lines) and not lines[e][1].startswith(("s_cbranch", "s_branch", "s_endpgm")):
                            blk.append(e)
                            e += 1
                        mb = re.match(r"(?:s_cbranch_\w+|s_branch)\s+(\d+)", lines[e][1]) if e < len(lines) else None
                        has_req = any(lines[x][1].split()[0].startswith(("buffer_load", "global_load")) for x in blk)
                        if mb and has_req and int(mb.group(1)) >= 32768:
                            back = index.get(lines[e][0] + 4 + (int(mb.group(1)) - 65536) * 4)
                            if back is not None and i < back <= i + 4:
                                seq += blk
                                i = back
                                continue
                i += 1
            seq += list(range(lo, hi + 1)) + list(range(lo, hi + 1))
            if linear_tail:
                seq += list(range(hi + 1, len(lines)))
            if os.environ.get("ISA_DEBUG") and os.environ["ISA_DEBUG"] in fn:
                for x in seq:
                    t0 = lines[x][1]
                    if t0.split()[0].startswith(("buffer_", "global_", "s_waitcnt", "s_cbranch", "s_branch")):
                        print("   ", hex(lines[x][0]), t0[:90])
            for (a, t, r) in replay(lines, seq, weakest_of_alternatives=any(k in fn for k in ALTERNATIVE_WAITS))[:4]:
                reports.append((fn, hex(a), t, r))
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
    return seen, reports


if __name__ == "__main__":
    if not L.OBJDUMP or not os.path.exists(L.LIB):
        print("isa_asmloads: cannot check (no llvm-objdump or no library)")
        sys.exit(2)
    seen, reports = check()
    if seen == 0:
        print("isa_asmloads: no gemm_wn_mma_kernel found — nothing was checked")
        sys.exit(2)
    names = sorted(set(r[0] for r in reports))
    dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.strip().split("\n"))) if names else {}
    print(f"isa_asmloads: {seen} MFMA tile kernels replayed; destination of an outstanding load touched early in {len(names)} kernels")
    for (fn, a, t, r) in reports[:40]:
        print(f"  EARLY {dem.get(fn, fn)[:120]} @{a}: {t[:80]}  (v{r})")
    sys.exit(1 if reports else 0)
