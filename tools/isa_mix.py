#!/usr/bin/env python3
"""Static instruction mix of the wide kernels, weighted by MEASURED issue costs.

  isa_mix.py [--json out.json] [--rates profiles/r03_valu_rate.txt] [kernel substrings...]

Compiles kernels_enc.hip / kernels_dec.hip to gfx950 assembly (hipcc -S, no GPU needed),
and for every selected kernel reports
  * the whole kernel: instructions by class;
  * every loop (the compiler's "Loop Header: Depth=n" annotations): instructions, VALU by
    issue class, the issue cycles of ONE wave-iteration = sum(class cost), and the top
    opcodes of the slow class.
Issue classes (tools/micro/valu_rate.hip, shader cycles per wave64 instruction and SIMD at
8 waves per SIMD; profiles/r03_valu_rate.txt):
  fast  ~2.2  VOP1 / VOP2 encodings with VGPR or inline-constant operands (v_add_u32,
              v_and_b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, v_add_u16 ...)
  slow  ~4.1  everything else on the vector ALU: every VOP3 (v_perm_b32, v_bfe_u32,
              v_lshl_add_u32, v_add3_u32, 64-bit shifts ...), packed 16-bit (v_pk_*), DPP,
              SDWA, compares, v_cndmask, v_readlane, left shifts, multiplies, and any
              VOP2 with an SGPR / literal operand in src0.
The guide (MI355X_MICROARCH.md) prices every wave64 VALU instruction at 2 cycles (SIMD-32);
both readings are reported: cycles_guide = 2 x VALU, cycles_measured = class-weighted."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "himg_amd", "csrc")

FAST_OPS = {  # measured at ~2.2 cycles (VGPR / inline operands)
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
    "v_mov_b32", "v_add_u16", "v_sub_u16", "v_ashrrev_i16", "v_lshrrev_b16", "v_not_b32", "v_bfrev_b32",
    "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_cvt_f32_u32",
}


def load_rates(path):
    fast, slow = 2.23, 4.10
    if path and os.path.exists(path):
        vals = {}
        for line in open(path):
            m = re.match(r"(\S.*?)\s+W=1:.*W=8:\s+([0-9.]+) cyc", line)
            if m:
                vals[m.group(1).strip()] = float(m.group(2))
        if "v_add_u32" in vals:
            fast = vals["v_add_u32"]
        if "v_perm_b32" in vals:
            slow = vals["v_perm_b32"]
    return fast, slow


def classify(line):
    """-> (kind, opcode) with kind in fast / slow / salu / lds / vmem / other."""
    parts = line.split(None, 1)
    op = parts[0]
    args = parts[1] if len(parts) > 1 else ""
    if op.startswith("s_"):
        return "salu", op
    if op.startswith("ds_"):
        return "lds", op
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", op
    if not op.startswith("v_"):
        return "other", op
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith(("_e64", "_dpp", "_sdwa")) or base.startswith(("v_pk_", "v_cmp", "v_readlane", "v_readfirstlane",
                                                                  "v_writelane", "v_permlane")):
        return "slow", base
    if base in FAST_OPS:
        ops = [a.strip() for a in args.split(",")]
        src = ops[1:] if len(ops) > 1 else []
        # an SGPR, vcc / exec or a 32-bit literal as a source makes the VOP2 slow
        for a in src:
            if re.match(r"^(s\d+|s\[|vcc|exec|ttmp)", a):
                return "slow", base + " (sgpr operand)"
        return "fast", base
    return "slow", base


def compile_s(name):
    out = os.path.join("/tmp", "himg_isa_%s.s" % name.replace(".hip", ""))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S",
           os.path.join(CSRC, name), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def demangle(n):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
            if out:
                return out
        except OSError:
            pass
    return n


def analyse(body, fast_c, slow_c):
    lines = [l.rstrip() for l in body.splitlines()]
    insts = []      # (index in lines, kind, opcode)
    labels = {}
    loop_at = {}    # label -> depth
    pending_depth = None
    # Which loop a basic block belongs to comes from the compiler's own annotations ("in
    # Loop: Header=BBn_m", "This (Inner) Loop Header", "Parent Loop BBn_k"), which hold
    # whatever the layout (rotated loops, latches in front of their headers, cold blocks
    # moved away).  own[h] = instructions of the blocks whose INNERMOST loop is h.
    own = collections.defaultdict(list)
    parent, depth_of, names = {}, {}, collections.defaultdict(set)
    regions = {}    # straight-line regions: name -> [begin, end) in insts (HIMG_REGION_BEGIN / _END outside loops)
    cur = None      # innermost loop of the current block
    inst_loop = []  # innermost loop of every instruction
    begin_at, end_at = collections.defaultdict(list), collections.defaultdict(list)   # marker name -> positions in insts
    for i, l in enumerate(lines):
        s = l.strip()
        blk = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)", s)
        if blk:
            ann = " ".join(lines[i:i + 8 if blk.group(1) else i + 1])
            ann = ann.split("\n")[0]
            # only the comment lines that directly follow the label belong to it
            k, ann = i + 1, lines[i]
            while k < len(lines) and lines[k].strip().startswith(";") and not lines[k].strip().startswith("; %bb.") and "HIMG_" not in lines[k] and "#ASM" not in lines[k]:
                ann += " " + lines[k]
                k += 1
            if blk.group(1) and "Loop Header: Depth=" in ann:
                cur = blk.group(1)
                depth_of[cur] = int(re.search(r"Loop Header: Depth=(\d+)", ann).group(1))
                ps = re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", ann)
                parent[cur] = max(ps, key=lambda t: int(t[1]))[0] if ps else None
            else:
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", ann)
                cur = h.group(1) if h else None
        sp = re.match(r"^;\s*HIMG_SPAN_(BEGIN|END)\s+(\S+)", s)
        if sp:   # a straight-line span, inside a loop or not
            if sp.group(1) == "BEGIN":
                regions.setdefault(sp.group(2), [len(insts), None])
            elif sp.group(2) in regions:
                regions[sp.group(2)][1] = len(insts)
            continue
        mk = re.match(r"^;\s*HIMG_REGION_(BEGIN|END)\s+(\S+)", s)
        if mk:
            (begin_at if mk.group(1) == "BEGIN" else end_at)[mk.group(2)].append(len(insts))
            if cur is not None:
                if mk.group(1) == "BEGIN":
                    names[cur].add(mk.group(2))
            elif mk.group(1) == "BEGIN":
                regions.setdefault(mk.group(2), [len(insts), None])
            elif mk.group(2) in regions:
                regions[mk.group(2)][1] = len(insts)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            # the annotation follows the label on the next comment lines
            for k in range(i, min(i + 6, len(lines))):
                d = re.search(r"Loop Header: Depth=(\d+)", lines[k])
                if d:
                    loop_at[m.group(1)] = int(d.group(1))
                    break
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        kind, op = classify(s)
        insts.append((kind, op, s))
        inst_loop.append(cur)
        if cur is not None:
            own[cur].append(insts[-1])
    # loops: a backward branch to a loop header label
    loops = []
    for pos, (kind, op, s) in enumerate(insts):
        m = re.search(r"(\.LBB\d+_\d+)$", s)
        if kind == "salu" and op.startswith(("s_cbranch", "s_branch")) and m and m.group(1) in labels and labels[m.group(1)] <= pos \
                and m.group(1) in loop_at:
            loops.append((m.group(1), labels[m.group(1)], pos + 1))
    def mix(seq):
        c = collections.Counter(k for k, _, _ in seq)
        slow_ops = collections.Counter(o for k, o, _ in seq if k == "slow")
        valu = c["fast"] + c["slow"]
        return {"instructions": len(seq), "valu": valu, "valu_fast": c["fast"], "valu_slow": c["slow"],
                "salu": c["salu"], "lds": c["lds"], "vmem": c["vmem"],
                "valu_cycles_guide": 2 * valu,
                "valu_cycles_measured": round(c["fast"] * fast_c + c["slow"] * slow_c, 1),
                "mean_cost_per_valu": round((c["fast"] * fast_c + c["slow"] * slow_c) / valu, 3) if valu else None,
                "top_slow": slow_ops.most_common(5)}
    res = {"kernel": mix(insts), "loops": [], "regions": {}}
    seen = set()
    uniq = []
    for lab, a, b in sorted(loops, key=lambda t: (t[1], -t[2])):
        if (lab, a) in seen:
            continue
        seen.add((lab, a))
        uniq.append((lab, a, b))
    for lab, a, b in uniq:
        m_ = mix(insts[a:b])
        if m_["instructions"] >= 12:
            m_.update({"header": lab, "depth": loop_at[lab]})
            res["loops"].append(m_)
    # Named loops (HIMG_REGION_BEGIN inside a loop body names that loop): the mix of the
    # loop's OWN blocks -- its nested loops are listed on their own -- for
    # tools/dynamic_mix.py, which weights them with measured trip counts.
    # The HOT PATH of a named loop: from its BEGIN comment along the layout -- an
    # unconditional branch is followed; `s_cbranch_execz` to a block of the same loop nest
    # is TAKEN (it skips a side path no lane needs: the path is the one every iteration
    # executes at least); every other conditional branch falls through -- to its END comment.
    def inside(lp, h):   # is loop lp == h or nested in h
        while lp is not None:
            if lp == h:
                return True
            lp = parent.get(lp)
        return False
    def hot_path(name, start, h):
        ends = set(end_at.get(name, []))
        seq, pos, steps = [], start, 0
        while pos < len(insts) and steps < 4000:
            if pos in ends and seq:
                return seq
            kind, op, txt = insts[pos]
            seq.append(insts[pos])
            steps += 1
            tgt = re.search(r"\.L(BB\d+_\d+)$", txt)
            if kind == "salu" and tgt and ("." + "L" + tgt.group(1)) in labels:
                tpos = labels[".L" + tgt.group(1)]
                tin = tpos < len(inst_loop) and inside(inst_loop[tpos], h)
                if op.startswith("s_branch") or (op.startswith("s_cbranch_execz") and tin):
                    pos = tpos
                    continue
            pos += 1
        return None
    res["named_loops"] = []
    for h, nm in names.items():
        kids = [c for c, p_ in parent.items() if p_ == h]
        entry = dict(mix(own[h]), names=sorted(nm), header=h, depth=depth_of.get(h), children=kids, parent=parent.get(h))
        for n in nm:
            for b in begin_at[n]:
                if b < len(inst_loop) and inst_loop[b] == h:
                    hp = hot_path(n, b, h)
                    if hp:
                        entry["hot_path"] = mix(hp)
        res["named_loops"].append(entry)
    # (compact: only what has instructions worth a line)
    res["all_loops_own"] = {h: {"instructions": len(v), "valu": sum(1 for k, _, _ in v if k in ("fast", "slow")),
                                "depth": depth_of.get(h), "parent": parent.get(h)}
                            for h, v in own.items() if len(v) >= 40}
    for name, (a, b) in regions.items():
        if b is not None and b > a:
            res["regions"][name] = mix(insts[a:b])
    return res


def main():
    args = sys.argv[1:]
    out_json, rates = None, os.path.join(ROOT, "profiles", "r03_valu_rate.txt")
    while args and args[0].startswith("--"):
        if args[0] == "--json":
            out_json = args[1]
        elif args[0] == "--rates":
            rates = args[1]
        args = args[2:]
    want = args or ["k_frontILb1ELi512", "5k_tokENS", "k_emit_tokILi8", "k_pix_fwdILb1ELi512ELb1", "k_tok_hist", "k_emit_tILi",
                    "k_row_countILb1", "k_row_count_w", "k_dec_row_fusedILi512", "k_lowres_avg"]
    fast_c, slow_c = load_rates(rates)
    report = {"issue_cycles": {"fast": fast_c, "slow": slow_c, "guide": 2.0,
                               "source": os.path.relpath(rates, ROOT) if os.path.exists(rates) else "defaults"},
              "kernels": {}}
    for src in ("kernels_enc.hip", "kernels_dec.hip"):
        s = compile_s(src)
        funcs = re.split(r"\n(_ZN8himg_dev[^\n:]+):", s)
        for i in range(1, len(funcs), 2):
            name, body = funcs[i], funcs[i + 1].split(".Lfunc_end")[0]
            if not any(p in name for p in want):
                continue
            short = demangle(name).replace("himg_dev::", "").split("(")[0].replace("void ", "")
            report["kernels"][short] = analyse(body, fast_c, slow_c)
    for k, v in report["kernels"].items():
        kk = v["kernel"]
        print("%-34s %5d instr, VALU %4d (fast %4d, slow %4d), mean %.2f cyc/VALU, SALU %4d, LDS %3d, VMEM %3d"
              % (k, kk["instructions"], kk["valu"], kk["valu_fast"], kk["valu_slow"], kk["mean_cost_per_valu"] or 0,
                 kk["salu"], kk["lds"], kk["vmem"]))
        for lp in sorted(v["loops"], key=lambda l: -l["instructions"])[:6]:
            print("    loop %-12s depth %d: %4d instr, VALU %3d (slow %3d) = %6.1f cyc measured / %4d guide; slow: %s"
                  % (lp["header"], lp["depth"], lp["instructions"], lp["valu"], lp["valu_slow"],
                     lp["valu_cycles_measured"], lp["valu_cycles_guide"],
                     ", ".join("%s x%d" % (o, n) for o, n in lp["top_slow"])))
    if out_json:
        json.dump(report, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
