#!/usr/bin/env python3
"""In-flight register check for the kernels that issue their own vector-memory loads from inline asm (round-4 advice).

conv3x3_fat / conv1x1_fat / conv1x1_duo / bottleneck_seam keep the destinations of asm `global_load_dwordx4` instructions (the weight
ring, residual quads) in compiler-visible VGPRs and wait for them with hand-counted `s_waitcnt vmcnt(N)`. hipcc does not know that a
load is in flight: if it copied (`v_mov`), re-used or spilled such a register between the load and the wait that covers it, the kernel
would compute on garbage without any tool noticing. This script replays a kernel's gfx950 assembly (hipcc -save-temps) in program
order:

  * every instruction counted by vmcnt (global / buffer / flat loads and stores, LDS-DMA loads, atomics) enters a FIFO, loads with their
    destination VGPRs; the counter retires IN ORDER (what the kernels' own counted waits assume);
  * `s_waitcnt vmcnt(N)` retires all but the youngest N entries;
  * any other instruction that names (reads or writes) a VGPR of a still-pending load is a violation;
  * control flow: one execution path is walked -- unconditional branches are followed, every loop runs three iterations (first,
    steady state, last) and is then left (by its backward branch falling through, or by a forward branch that jumps past the back-edge),
    the optional side of a forward diamond inside a body is always executed; instructions the walk never reached are counted in the report.

usage: ring_hazard_check.py FILE.s KERNEL_SYMBOL_REGEX   -> one line per kernel; exit 1 on a violation or when nothing matched."""
import re
import sys

VM_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load)_")
VM_OTHER = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic|flat_atomic)_")
VREG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+)\b)")
WAIT_VM = re.compile(r"vmcnt\((\d+)\)")


def vregs(text):
    regs = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            regs.add(int(m.group(3)))
        else:
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def parse(body):
    """-> list of (kind, payload, text): kind in label / branch / inst"""
    prog = []
    for raw in body:
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        s = line.strip()
        if s.endswith(":") and not s.startswith("."):
            continue
        if re.match(r"^\.L[A-Za-z0-9_$.]+:$", s):
            prog.append(("label", s[:-1], s))
            continue
        if s.startswith(".") or s.startswith("#"):
            continue
        mnem = s.split()[0]
        if mnem in ("s_branch",) or mnem.startswith("s_cbranch"):
            prog.append(("branch", s.split()[-1], s))
        else:
            prog.append(("inst", mnem, s))
    return prog


def check(prog):
    labels = {p[1]: i for i, p in enumerate(prog) if p[0] == "label"}
    fifo = []          # entries: (index, set of destination VGPRs)
    violations = []
    stats = {"asm_loads": 0, "waits": 0, "max_pending": 0}

    def step(i):
        kind, mnem, text = prog[i]
        if kind != "inst":
            return
        if mnem == "s_waitcnt":
            m = WAIT_VM.search(text)
            if m:
                stats["waits"] += 1
                n = int(m.group(1))
                if len(fifo) > n:
                    del fifo[: len(fifo) - n]
            return
        ops = text[len(mnem):]
        pending = set().union(*[e[1] for e in fifo]) if fifo else set()
        if VM_LOAD.match(mnem):
            first = ops.split(",")[0]
            lds = "lds" in mnem
            dst = set() if lds else vregs(first)
            rest = ops if lds else ",".join(ops.split(",")[1:])
            touched = (vregs(rest) | dst) & pending
            if touched:
                violations.append((i, text, sorted(touched)))
            fifo.append((i, dst))
            stats["asm_loads"] += 1
            stats["max_pending"] = max(stats["max_pending"], len(fifo))
            return
        touched = vregs(ops) & pending
        if touched:
            violations.append((i, text, sorted(touched)))
        if VM_OTHER.match(mnem):
            fifo.append((i, set()))
            stats["max_pending"] = max(stats["max_pending"], len(fifo))

    # walk ONE execution path: unconditional branches are followed; a conditional branch goes its loop-friendly way on its first two
    # visits (backward: taken, forward: not taken -- the skipped side of a diamond is a subset of the fall-through) and the other
    # way on the third, so that every loop runs three iterations (first, steady state, last) and is then left
    visits = {}
    seen = set()
    back_edges = [(k, labels[p_[1]]) for k, p_ in enumerate(prog) if p_[0] == "branch" and p_[1] in labels and labels[p_[1]] < k]
    i, steps = 0, 0
    while i < len(prog) and steps < 50 * len(prog):
        steps += 1
        kind, payload, text = prog[i]
        seen.add(i)
        if kind == "branch" and payload in labels:
            target = labels[payload]
            if text.split()[0] == "s_branch":
                i = target
                continue
            n = visits[i] = visits.get(i, 0) + 1
            backward = target < i
            if backward:
                take = n % 3 != 0
            else:
                # a forward conditional branch is followed only where it LEAVES a loop (a back-edge lies between it and its target and
                # closes a loop that contains it), and then on its third visit; a forward branch that merely skips code inside the loop body
                # is never followed: skipping one optional part but not the next can be a path no execution takes
                leaves = any(i < b < target and h <= i for b, h in back_edges)
                take = leaves and n % 3 == 0
            if take:
                i = target
                continue
        elif kind == "inst":
            if payload == "s_endpgm":
                break
            step(i)
        i += 1
    stats["unvisited"] = sum(1 for k, p in enumerate(prog) if p[0] == "inst" and k not in seen)
    return violations, stats


def main():
    path, pattern = sys.argv[1], re.compile(sys.argv[2])
    lines = open(path).read().splitlines()
    found, bad = 0, 0
    i = 0
    while i < len(lines):
        m = re.match(r"^([A-Za-z_][A-Za-z0-9_$.]*):", lines[i])
        if m and pattern.search(m.group(1)) and not m.group(1).startswith(".L"):
            name = m.group(1)
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            violations, stats = check(parse(lines[i + 1:j]))
            seen = set()
            uniq = [v for v in violations if not (v[1] in seen or seen.add(v[1]))]
            found += 1
            print("%s: %d vector-memory loads and %d vmcnt waits walked (%d instructions not reached), deepest queue %d, in-flight register touched: %d" % (
                name, stats["asm_loads"], stats["waits"], stats["unvisited"], stats["max_pending"], len(uniq)))
            for _, text, regs in uniq[:8]:
                print("    %s   <- pending v%s" % (text, regs))
            bad += len(uniq)
            i = j
        i += 1
    if not found:
        print("no kernel matching %s in %s -- nothing was checked" % (sys.argv[2], path))
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
