#!/usr/bin/env python3
"""Find s_waitcnt vmcnt(..) instructions that no nearby instruction seems to need.

The compiler places s_waitcnt from a forward dataflow over the (structurised) control
flow graph: a load that is still in flight when some code path is LEFT stays pending on
its destination register along every static path from there -- dynamically impossible
ones included -- and the first write to that register in a later loop then carries a
wait on every iteration (round 5: the row kernel's write loop waited for its own
prefetch at every step because of the token-tail loop's unused last prefetch).

Usage: waitcnt_audit.py file.s [kernel-name-substring]
For every wait inside a loop body it looks at the instruction right behind it (the compiler
puts a wait directly in front of what needs it): when that instruction READS no register
that any VMEM load of the function writes (a heuristic: register names, not control flow),
the wait protects a WRITE to a register the dataflow still holds pending -- printed.
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)\w*\s+(v\[?\d+(?::\d+)?\]?)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit(lines, name):
    ever = set()        # VGPRs that are the destination of a VMEM load anywhere in the function
    for ln in lines:
        m = VMEM.match(ln)
        if m:
            ever |= regs(m.group(2))
    loaded = set()      # VGPRs written by a VMEM load and not yet overwritten by anything else (text order)
    depth = 0
    n_sus = 0
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        if "Loop Header" in s or "Inner Loop Header" in s:
            m = re.search(r"Depth=(\d+)", s)
            depth = int(m.group(1)) if m else 1
        if s.startswith(".LBB") and "Loop" not in s and "in Loop" not in lines[i + 1 if i + 1 < len(lines) else i]:
            pass
        m = VMEM.match(ln)
        if m:
            loaded |= regs(m.group(2))
            i += 1
            continue
        if s.startswith("s_waitcnt") and "vmcnt" in s and "in Loop" in "".join(lines[max(0, i - 40):i + 1]) + "":
            # the stretch behind the wait
            j = i + 1
            reads, writes, stretch = set(), set(), []
            while j < len(lines) and len(stretch) < 1:   # the compiler puts the wait right in front of the instruction that needs it
                t = lines[j].strip()
                if not t or t.startswith(";"):
                    j += 1
                    continue
                if t.startswith(".LBB") or t.startswith("s_cbranch") or t.startswith("s_branch"):
                    break
                stretch.append(t)
                ops = t.split(None, 1)
                if len(ops) == 2:
                    parts = ops[1].split(",")
                    dst = parts[0]
                    is_store = ops[0].startswith(("global_store", "buffer_store", "ds_write", "ds_or", "ds_add", "global_atomic", "flat_store"))
                    if is_store:
                        reads |= regs(ops[1])
                    else:
                        writes |= regs(dst)
                        reads |= regs(",".join(parts[1:]))
                j += 1
            near = set()     # load destinations within 150 lines either side (the loop and its surroundings)
            for t in lines[max(0, i - 150):i + 150]:
                mm = VMEM.match(t)
                if mm:
                    near |= regs(mm.group(2))
            if not (reads & near):
                n_sus += 1
                hit = sorted(writes & loaded)
                print("%s: line %d: %s  -- no loaded register read behind it%s" % (
                    name, i + 1, s, (" (writes pending v%s)" % hit) if hit else ""))
                for t in stretch[:6]:
                    print("      " + t)
        # any other instruction that writes VGPRs clears their "loaded" mark
        ops = s.split(None, 1)
        if len(ops) == 2 and not s.startswith((";", ".")):
            dst = ops[1].split(",")[0]
            if not ops[0].startswith(("global_store", "buffer_store", "ds_write", "ds_or", "s_")):
                loaded -= regs(dst)
        i += 1
    return n_sus


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    txt = open(path).read().split("\n")
    # split into functions
    start = None
    name = None
    total = 0
    for k, ln in enumerate(txt):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            start, name = k, m.group(1)
        if ln.startswith(".Lfunc_end") and start is not None:
            if want in name:
                total += audit(txt[start:k], name[:60])
            start = None
    print("suspicious waits: %d" % total)


if __name__ == "__main__":
    main()
