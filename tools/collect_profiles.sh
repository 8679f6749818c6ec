#!/bin/bash
# Copy what tools/refresh_profiles.sh left under gpurun_out/<dir> into profiles/ under
# this round's names:  tools/collect_profiles.sh gpurun_out/r03c r03
set -eu
SRC=$1
TAG=$2
grep '^{' "$SRC/bench.json" | tail -1 > "profiles/${TAG}_bench.json"
grep '^{' "$SRC/bench_b64.json" | tail -1 > "profiles/${TAG}_bench_b64.json"
cp "$SRC/cfg3_sweep.json" "profiles/${TAG}_cfg3_sweep.json"
cp "$SRC/cfg4_rows.json" "profiles/${TAG}_cfg4_rows.json"
cp "$SRC/latency.json" "profiles/${TAG}_latency.json"
python3 - "$SRC/configs.jsonl" "profiles/${TAG}_configs.json" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
out = [{"workload": r["config"]["workload"], "bit_exact": r["config"]["bit_exact"], "value": r["value"], "encode_mpx_s": r["encode_mpx_s"], "decode_mpx_s": r["decode_mpx_s"], "ms_per_step": r["ms_per_step"]} for r in rows]
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
cp "$SRC"/stats_default/*/*_kernel_stats.csv "profiles/${TAG}_kernel_stats_default.csv"
cp "$SRC"/stats_b64/*/*_kernel_stats.csv "profiles/${TAG}_kernel_stats_b64.csv"
cp "$SRC/pmc/summary.json" "profiles/${TAG}_pmc_summary.json"
cp "$SRC/traffic.json" profiles/traffic.json
[ -f "$SRC/calib/calibration.json" ] && cp "$SRC/calib/calibration.json" "profiles/${TAG}_calibration.json"
[ -f "$SRC/graph_latency.json" ] && cp "$SRC/graph_latency.json" "profiles/${TAG}_graph_latency.json"
[ -f "$SRC/isa_mix.json" ] && cp "$SRC/isa_mix.json" "profiles/${TAG}_isa_mix.json"
[ -f "$SRC/loop_counts.json" ] && cp "$SRC/loop_counts.json" "profiles/${TAG}_loop_counts.json"
[ -f "$SRC/dynamic_mix.json" ] && cp "$SRC/dynamic_mix.json" "profiles/${TAG}_dynamic_mix.json"
ls -la profiles/${TAG}_* profiles/traffic.json
