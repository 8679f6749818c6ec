#!/usr/bin/env python3
"""Dynamic instruction mix of the hottest loops: measured trip counts x the loops' static mix.

The issue-rate reading of a kernel (bench.py `roofline_valu.issue_frac_measured_mix`) prices its
SQ_INSTS_VALU count at the class-weighted mean cost of its instructions.  Taken over the
kernel's STATIC mix that mean includes every cold path; here it is taken over what the kernel
runs:

  dynamic_mix.py --run counts.json [frames]     (GPU box; HIMG_EXTRA_HIPCC_FLAGS=-DHIMG_LOOP_COUNTS)
      rebuilds the library with the loop counters of csrc/loop_counts.h compiled in, encodes
      and decodes `frames` 4096x4096 q50 randtile frames through the device API (every stream
      and picture checked against the golden table) and writes, per marked loop, the iterations the
      wavefronts executed and the sum over their lanes.
  dynamic_mix.py --analyse counts.json [out.json]    (anywhere: hipcc -S, no GPU)
      per kernel: VALU instructions per frame from PMC (profiles/traffic.json), split into
        * marked straight-line regions (executed once per wavefront: waves x static count),
        * marked loops: wave-iterations x the loop's HOT PATH (tools/isa_mix.py: the path
          every iteration takes, side paths that `s_cbranch_execz` skips left out),
        * the rest (side paths inside the loops, prologues, unmarked loops) = PMC minus the
          two, priced at the mix of the kernel's remaining static instructions;
      and from that the class-weighted cost per executed VALU instruction and the issue
      fraction of the kernel's duration (1024 SIMDs at the clock rocprof saw).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

W, H, Q = 4096, 4096, 50
SYMBOLS = W * H * 4
ROWS = H // 8

# marked loop / region -> (kernel as tools/isa_mix.py names it, counter side, counter index or None)
LOOPS = {
    "dec.write": ("k_dec_row_fused<512>", "dec", 0),
    "dec.count": ("k_row_count_w", "dec", 1),
    # (round 6: batches tokenise with k_tok and pack with k_emit_tok; k_emit_t's loops -- enc.iter /
    # enc.walk / enc.stage, counters 0..2 -- only run for single frames and LRES spans now)
    "tokr.iter": ("k_tok", "enc", 3),
    "tokr.walk": ("k_tok", "enc", 4),
    "tok.iter": ("k_emit_tok<8>", "enc", 5),
}
# an outer loop's hot path passes through its inner loops (once per inlined copy): taken out of it
INNER = {"enc.iter": [("enc.walk", 1), ("enc.stage", 1)]}
OWN_BLOCKS = {"tokr.iter", "tokr.walk"}
REGIONS = {"dec.transform": ("k_dec_row_fused<512>", 16 * ROWS)}   # wavefronts per frame that run it once
WAVES = {"k_dec_row_fused<512>": 16 * ROWS, "k_row_count_w": ROWS, "k_tok": 4 * ROWS, "k_emit_tok<8>": ROWS}


def run(out_path, frames):
    if "HIMG_LOOP_COUNTS" not in os.environ.get("HIMG_EXTRA_HIPCC_FLAGS", ""):
        sys.exit("set HIMG_EXTRA_HIPCC_FLAGS=-DHIMG_LOOP_COUNTS (the product build has no counters)")
    import numpy as np
    import torch
    from himg_amd import build
    build.build_lib()   # (the flags differ from the stamp of the library in the tree: rebuilt)
    import himg_amd
    eng = himg_amd.Engine(0)
    imgs = [himg_amd.synth("randtile", s, W, H) for s in range(frames)]
    d_frames = torch.from_numpy(np.stack(imgs)).cuda()
    cap = (himg_amd.max_packed_size(W, H, 4) + 255) // 256 * 256
    d_out = torch.zeros((frames, cap), dtype=torch.uint8, device="cuda")
    d_sizes = torch.zeros(frames, dtype=torch.int32, device="cuda")
    d_st = torch.zeros(frames, dtype=torch.int32, device="cuda")
    d_pix = torch.empty((frames, H, W, 4), dtype=torch.uint8, device="cuda")

    def step():
        eng.encode_device(d_frames, frames, W, H, 4, 4, Q, True, d_out, cap, d_sizes, d_st, 0)
        torch.cuda.synchronize()
        sizes = d_sizes.cpu().numpy().astype(np.uint32)
        eng.decode_device(d_out, cap, sizes, frames, W, H, 4, d_pix, d_st, 0)
        torch.cuda.synchronize()
        return sizes
    step()
    eng.debug_read("loop_counts", 0, 128, np.uint64)                   # reset
    eng.debug_read("loop_counts", 0, 128, np.uint64, decoder=True)
    sizes = step()
    enc = eng.debug_read("loop_counts", 0, 128, np.uint64).reshape(8, 2)
    dec = eng.debug_read("loop_counts", 0, 128, np.uint64, decoder=True).reshape(8, 2)
    # the instrumented build computes what the product build computes: every frame against the
    # golden table of the batch workload (tests/golden/batch_4096x4096_q50.json)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "batch_4096x4096_q50.json")))["seeds"]
    for i in range(frames):
        size, sfnv, pfnv = gold[i]
        assert int(sizes[i]) == size, "frame %d: packed size" % i
        assert himg_amd.fnv1a64(d_out[i, :size].cpu().numpy()) == sfnv, "frame %d: stream differs from the golden table" % i
        assert himg_amd.fnv1a64(d_pix[i].cpu().numpy().ravel()) == pfnv, "frame %d: pixels differ from the golden table" % i
    json.dump({"frames": frames, "workload": "%dx%d RGBA randtile q=%d, seeds 0..%d" % (W, H, Q, frames - 1),
               "enc": enc.tolist(), "dec": dec.tolist(),
               "columns": ["wavefront iterations", "lane iterations"]}, open(out_path, "w"), indent=1)
    print(open(out_path).read())


def analyse(counts_path, out_path):
    import isa_mix
    counts = json.load(open(counts_path))
    frames = counts["frames"]
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    fast_c, slow_c = isa_mix.load_rates(os.path.join(ROOT, "profiles", "r03_valu_rate.txt"))
    static = {}
    for src in ("kernels_enc.hip", "kernels_dec.hip"):
        s = isa_mix.compile_s(src)
        import re
        funcs = re.split(r"\n(_ZN8himg_dev[^\n:]+):", s)
        for i in range(1, len(funcs), 2):
            short = isa_mix.demangle(funcs[i]).replace("himg_dev::", "").split("(")[0].replace("void ", "")
            if short in WAVES:
                static[short] = isa_mix.analyse(funcs[i + 1].split(".Lfunc_end")[0], fast_c, slow_c)

    def cost(m):
        return m["valu_fast"] * fast_c + m["valu_slow"] * slow_c
    out = {"note": __doc__.split("\n\n")[1].strip(), "counts": counts, "issue_cycles": {"fast": fast_c, "slow": slow_c},
           "kernels": {}}
    for kern, waves in WAVES.items():
        st = static[kern]
        pmc = traffic["valu"][kern]
        total = pmc["wave_insts_per_symbol"] * SYMBOLS            # VALU wave-instructions per frame
        parts, used_valu, used_cost, used_fast, used_slow = [], 0.0, 0.0, 0, 0
        for name, (k2, w_once) in REGIONS.items():
            if k2 != kern or name not in st["regions"]:
                continue
            m = st["regions"][name]
            v = w_once * m["valu"]
            parts.append({"part": name + " (straight-line, once per wavefront and row)", "valu_static": m["valu"],
                          "valu_per_frame": v, "mean_cost": m["mean_cost_per_valu"]})
            used_valu += v
            used_cost += w_once * cost(m)
            used_fast += m["valu_fast"]; used_slow += m["valu_slow"]
        inner_once = {}   # the hot path of an outer loop passes through one iteration of its inner loops
        for name, (k2, side, idx) in LOOPS.items():
            if k2 != kern:
                continue
            # (tokr.*: the compiler schedules ALU instructions across the marker comments of these small
            # loops -- the marked span holds 8 of the walk's 19 instructions -- so their path is the
            # loop's OWN blocks: nested loops are listed on their own, the side paths are a few instructions)
            if name in OWN_BLOCKS:
                cands = [dict(l, hot_path={k_: l[k_] for k_ in ("valu", "valu_fast", "valu_slow", "instructions", "salu", "lds", "vmem")})
                         for l in st["named_loops"] if name in l["names"]]
            else:
                cands = [l for l in st["named_loops"] if name in l["names"] and l.get("hot_path")]
            if not cands:
                continue
            lp = cands[0]
            # (the compiler clones a loop that is inlined in several places: the counter adds them up,
            # the path is their mean)
            hp = dict(lp["hot_path"])
            for key in ("valu", "valu_fast", "valu_slow", "instructions", "salu", "lds", "vmem"):
                hp[key] = sum(c["hot_path"][key] for c in cands) / len(cands)
            hp["mean_cost_per_valu"] = round((hp["valu_fast"] * fast_c + hp["valu_slow"] * slow_c) / hp["valu"], 3)
            wave_it = counts[side][idx][0] / frames
            lane_it = counts[side][idx][1] / frames
            inner_once[name] = hp
            parts.append({"part": name + " (loop, hot path)", "valu_static": hp["valu"], "hot_path": hp,
                          "own_blocks_valu": lp["valu"], "wave_iterations_per_frame": round(wave_it, 1),
                          "lane_iterations_per_frame": round(lane_it, 1),
                          "active_lanes_per_wave_iteration": round(lane_it / wave_it, 2) if wave_it else None,
                          "iterations_per_wavefront": round(wave_it / waves, 2),
                          "mean_cost": hp["mean_cost_per_valu"]})
        # an outer loop's path contains one pass of each copy of its inner loops: take them out of it
        for p in parts:
            outer = p["part"].split(" ")[0]
            if outer in INNER and "hot_path" in p:
                for inner, copies in INNER[outer]:
                    if inner in inner_once:
                        for key in ("valu", "valu_fast", "valu_slow"):
                            p["hot_path"][key] = max(p["hot_path"][key] - copies * inner_once[inner][key], 0)
                p["valu_static"] = p["hot_path"]["valu"]
                p["mean_cost"] = round(cost(p["hot_path"]) / p["hot_path"]["valu"], 3) if p["hot_path"]["valu"] else None
        for p in parts:
            if "hot_path" not in p:
                continue
            v = p["wave_iterations_per_frame"] * p["hot_path"]["valu"]
            p["valu_per_frame"] = v
            used_valu += v
            used_cost += p["wave_iterations_per_frame"] * cost(p["hot_path"])
            used_fast += p["hot_path"]["valu_fast"]; used_slow += p["hot_path"]["valu_slow"]
        kk = st["kernel"]
        rest_fast, rest_slow = max(kk["valu_fast"] - used_fast, 0), max(kk["valu_slow"] - used_slow, 1)
        rest_mean = (rest_fast * fast_c + rest_slow * slow_c) / (rest_fast + rest_slow)
        rest = total - used_valu
        parts.append({"part": "rest: side paths inside the loops (refills, long codes, runs), prologue, unmarked loops "
                              "= PMC - the above, priced at the mix of the kernel's remaining static instructions",
                      "valu_per_frame": rest, "mean_cost": round(rest_mean, 3)})
        dyn_cost = used_cost + rest * rest_mean
        for p in parts:
            p["share_of_executed_valu"] = round(p["valu_per_frame"] / total, 3)
            p["valu_per_frame"] = round(p["valu_per_frame"])
            p.pop("hot_path", None)
        mean_dyn = dyn_cost / total
        simd_cycles = pmc["dur_us"] * 1e-6 * pmc["clock_GHz"] * 1e9 * 1024 / traffic["frames_per_launch"]   # per frame
        out["kernels"][kern] = {
            "valu_wave_instructions_per_frame_pmc": round(total), "wavefronts_per_frame": waves, "parts": parts,
            "mean_cost_per_valu_static_whole_kernel": kk["mean_cost_per_valu"],
            "mean_cost_per_valu_dynamic": round(mean_dyn, 3),
            "issue_frac_static_mix": round(total * kk["mean_cost_per_valu"] / simd_cycles, 3),
            "issue_frac_dynamic_mix": round(dyn_cost / simd_cycles, 3),
            "issue_frac_guide_2cyc": round(total * 2.0 / simd_cycles, 3)}
    js = json.dumps(out, indent=1)
    if out_path:
        open(out_path, "w").write(js)
    for k, v in out["kernels"].items():
        print("%-22s static %.2f -> dynamic %.2f cycles per VALU; issue fraction %.2f -> %.2f (guide's 2 cycles: %.2f)"
              % (k, v["mean_cost_per_valu_static_whole_kernel"], v["mean_cost_per_valu_dynamic"],
                 v["issue_frac_static_mix"], v["issue_frac_dynamic_mix"], v["issue_frac_guide_2cyc"]))
        for p in v["parts"]:
            print("    %5.1f %%  %s%s" % (100 * p["share_of_executed_valu"], p["part"][:70],
                                         "  (%s iterations per wavefront, %s lanes active)" % (p["iterations_per_wavefront"], p["active_lanes_per_wave_iteration"]) if "iterations_per_wavefront" in p else ""))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 16)
    elif len(sys.argv) >= 3 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        sys.exit(__doc__)
