#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call (from the repo root):
#   tools/refresh_profiles.sh <outdir under gpurun_out> [git sha]
# the bench line, kernel stats (default run and one stream), PMC passes + traffic,
# the config-3 occupancy sweep, the config-4 row profile, single-frame latency.
set -u
OUT=${1:-gpurun_out/refresh}
SHA=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
# The PMC passes first: bench.py quotes profiles/traffic.json (HBM bytes per launch, issue
# fractions) next to its live timings, so the file it reads must be this commit's.
# (At the DEFAULT batch, 128 frames per launch: what bench.py's line quotes is then read from
# these passes, not scaled from a smaller launch.)
(cd "$ROOT" && BENCH_ARGS="--streams 1 --no-rows" bash tools/profile_pmc.sh "$OUT/pmc" > "$ROOT/$OUT/pmc.log" 2>&1
 python3 tools/make_traffic_json.py "$OUT/pmc/summary.json" 128 "$OUT/traffic.json" "$SHA" && cp "$OUT/traffic.json" profiles/traffic.json)
python3 "$ROOT/bench.py" > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err"
# The same with 64 frames per launch: the batch DESIGN.md's per-kernel discussion is written for.
python3 "$ROOT/bench.py" --batch 64 --no-rows --no-extras --no-cpu-baseline > "$ROOT/$OUT/bench_b64.json" 2> "$ROOT/$OUT/bench_b64.err"
python3 "$ROOT/tools/occupancy_sweep.py" > "$ROOT/$OUT/cfg3_sweep.json" 2> "$ROOT/$OUT/cfg3_sweep.err"
python3 "$ROOT/tools/rows_profile.py" > "$ROOT/$OUT/cfg4_rows.json" 2> "$ROOT/$OUT/cfg4_rows.err"
python3 "$ROOT/tools/latency_profile.py" > "$ROOT/$OUT/latency.json" 2> "$ROOT/$OUT/latency.err"
# Other shapes of the same workload: config 5 (quality sweep), 2048^2 / 1024^2 / 1080p batches.
: > "$ROOT/$OUT/configs.jsonl"
for q in 10 30 70 90; do
  python3 "$ROOT/bench.py" --quality $q --batch 64 --no-rows --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 >> "$ROOT/$OUT/configs.jsonl"
done
for wh in "2048 2048" "1024 1024" "1920 1080"; do
  set -- $wh
  python3 "$ROOT/bench.py" --width $1 --height $2 --batch 256 --no-rows --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 >> "$ROOT/$OUT/configs.jsonl"
done
# Counter calibration on known byte counts (FETCH_SIZE / WRITE_SIZE factors, copy ceilings) and the
# single-frame chains replayed from a HIP graph.
bash "$ROOT/tools/calibrate_pmc.sh" "$OUT/calib" > "$ROOT/$OUT/calib.log" 2>&1
python3 "$ROOT/tools/graph_latency.py" > "$ROOT/$OUT/graph_latency.json" 2> "$ROOT/$OUT/graph_latency.err"
cd /tmp && export TMPDIR=/tmp
# The driver's command (default batch), then the 64-frame launch DESIGN.md's per-kernel tables are written for.
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_default" -- \
    python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-rows > "$ROOT/$OUT/stats_default.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_b64" -- \
    python3 "$ROOT/bench.py" --steps 5 --warmup 2 --batch 64 --no-cpu-baseline --no-extras --no-rows > "$ROOT/$OUT/stats_b64.log" 2>&1
cd "$ROOT"
# Static instruction mix of the wide kernels (no GPU needed, but this commit's sources), then the loops'
# measured trip counts from a build with the counters compiled in (the snapshot's library is rebuilt
# for it, and rebuilt again without them).
python3 tools/isa_mix.py --json "$OUT/isa_mix.json" > "$OUT/isa_mix.txt" 2>&1
HIMG_EXTRA_HIPCC_FLAGS=-DHIMG_LOOP_COUNTS python3 tools/dynamic_mix.py --run "$OUT/loop_counts.json" 16 > "$OUT/loop_counts.log" 2>&1
python3 -c "from himg_amd import build; build.build_lib()" > /dev/null 2>&1
python3 tools/dynamic_mix.py --analyse "$OUT/loop_counts.json" "$OUT/dynamic_mix.json" > "$OUT/dynamic_mix.txt" 2>&1
ls "$ROOT/$OUT"
