"""The measurement tools' view of the kernels: tools/isa_mix.py must find every loop and
region that csrc/loop_counts.h marks (the comments survive the optimiser, the loops keep a
hot path it can follow) -- tools/dynamic_mix.py's split of a kernel's executed instructions
rests on that.  CPU only: hipcc -S for gfx950, no GPU."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")),
                                reason="hipcc not available")


@pytest.fixture(scope="module")
def mixes():
    import re
    import isa_mix
    fast_c, slow_c = isa_mix.load_rates(os.path.join(ROOT, "profiles", "r03_valu_rate.txt"))
    out = {}
    for src in ("kernels_enc.hip", "kernels_dec.hip"):
        text = isa_mix.compile_s(src)
        funcs = re.split(r"\n(_ZN8himg_dev[^\n:]+):", text)
        for i in range(1, len(funcs), 2):
            short = isa_mix.demangle(funcs[i]).replace("himg_dev::", "").split("(")[0].replace("void ", "")
            if short in ("k_dec_row_fused<512>", "k_row_count_w", "k_emit_t<8, false>"):
                out[short] = isa_mix.analyse(funcs[i + 1].split(".Lfunc_end")[0], fast_c, slow_c)
    return out


@pytest.mark.parametrize("kernel,loop", [("k_dec_row_fused<512>", "dec.write"), ("k_row_count_w", "dec.count"),
                                         ("k_emit_t<8, false>", "enc.iter"), ("k_emit_t<8, false>", "enc.walk"),
                                         ("k_emit_t<8, false>", "enc.stage")])
def test_marked_loops_are_found_with_a_hot_path(mixes, kernel, loop):
    loops = [l for l in mixes[kernel]["named_loops"] if loop in l["names"] and l.get("hot_path")]
    assert loops, "no loop named %s with a hot path in %s" % (loop, kernel)
    hp = loops[0]["hot_path"]
    # a hot path is a handful of instructions, not the loop's cold blocks
    assert 5 <= hp["valu"] <= loops[0]["valu"] and hp["instructions"] < 400


def test_transform_region_is_found(mixes):
    r = mixes["k_dec_row_fused<512>"]["regions"].get("dec.transform")
    assert r and r["valu"] > 1000   # the straight-line transform of two planes


def test_issue_classes_cover_every_valu_instruction(mixes):
    for k, v in mixes.items():
        kk = v["kernel"]
        assert kk["valu"] == kk["valu_fast"] + kk["valu_slow"] and kk["valu"] > 1000, k
        assert 2.0 < kk["mean_cost_per_valu"] < 4.2, k
