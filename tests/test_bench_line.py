"""bench.py's contract, exercised the way the driver runs it: one JSON line on stdout
with the fields the harness reads, for N = 1 and -- launched by bench.py itself, two
ranks sharing the one GPU of the test box (--oversubscribe: gloo + CPU staging for the
collectives, not a measurement) -- for N = 2 in both modes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


def _run(args, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_help_lists_the_contract_flags():
    r = subprocess.run([sys.executable, BENCH, "--help"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--no-rows", "--rows-timeout"):
        assert flag in r.stdout


@pytest.mark.gpu
def test_single_gpu_line():
    d = _run(["--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-rows"])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "Mpixels/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 1000 and d["vs_baseline"] is None
    assert "4096x4096" in d["config"]["workload"] and "bit_exact" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3


@pytest.mark.gpu
def test_single_gpu_line_with_extras():
    """The default run's side measurements (copy ceilings incl. the hand-written calibration
    kernels, single-frame latency, host API) on a tiny batch: they must not take the line down."""
    d = _run(["--batch", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-rows"])
    ex = d["extras"]
    assert ex["hbm_copy_ceiling_GBs"] > 500
    assert ex["single_frame_latency_ms"]["encode"] > 0 and ex["single_frame_latency_ms"]["decode"] > 0
    if os.path.exists(os.path.join(ROOT, "himg_amd", "bin", "hbm_calib")):
        k = ex["hbm_ceiling_kernels_GBs"]
        assert k["read_only"] > 1000 and k["write_only"] > 1000 and k["copy_read_plus_write"] > 1000


@pytest.mark.gpu
def test_two_ranks_self_launched_frames():
    d = _run(["--gpus", "2", "--oversubscribe", "--batch", "2", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--no-extras", "--no-rows"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "2 rank" in d["config"]["parallelism"]


@pytest.mark.gpu
def test_two_ranks_self_launched_rows():
    d = _run(["--gpus", "2", "--oversubscribe", "--mode", "rows", "--width", "1024", "--height", "1024",
              "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"


@pytest.mark.gpu
def test_eight_ranks_self_launched_frames_and_rows():
    """`python bench.py --gpus 8` as the driver's 8-GPU box will run it -- the launcher, eight
    ranks, the seed split, the max-over-ranks reduction, then the `rows` leg (row-sharded encode
    and decode of one frame over the same eight ranks) -- executed once before that box is its
    first execution: all ranks share the one GPU here (--oversubscribe: gloo + CPU staging,
    not a measurement), one frame per rank and a 2048 x 2048 frame for the rows leg (256 block
    rows: sixteen-row shares for all eight ranks)."""
    d = _run(["--gpus", "8", "--oversubscribe", "--batch", "1", "--steps", "1", "--warmup", "1",
              "--no-cpu-baseline", "--no-extras", "--rows-size", "2048x2048", "--rows-timeout", "600"], timeout=1500)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 1
    assert "8 rank" in d["config"]["parallelism"]
    assert d["config"]["bit_exact"]          # every rank's frames checked against the golden table before timing
    rows = d["rows"]
    assert "error" not in rows, rows
    assert "2048x2048" in rows["workload"].replace(" ", "")
    assert rows["bit_exact"] == {"stream": "oracle", "pixels": "oracle"}, rows
    assert rows["encode_ms"]["mean"] > 0 and rows["decode_ms"]["mean"] > 0
