#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call (from the repo root):
#   tools/refresh_profiles.sh <outdir under gpurun_out>
# kernel stats (default run and one stream), PMC passes, the bench line itself.
set -u
OUT=${1:-gpurun_out/refresh}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
python3 "$ROOT/bench.py" > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_default" -- \
    python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$ROOT/$OUT/stats_default.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_streams1" -- \
    python3 "$ROOT/bench.py" --steps 5 --warmup 2 --streams 1 --batch 8 --no-cpu-baseline --no-extras > "$ROOT/$OUT/stats_streams1.log" 2>&1
cd "$ROOT" && BENCH_ARGS="--streams 1 --batch 8" bash tools/profile_pmc.sh "$OUT/pmc"
