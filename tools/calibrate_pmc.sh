#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of KNOWN byte counts (VERDICT r4 "do this" #3): tools/micro/hbm_calib
# under rocprofv3, one counter per pass; tools/summarize_calib.py divides the counters by the bytes
# the kernels are known to move and writes the factors profiles/ and summarize_pmc.py use.
#   tools/calibrate_pmc.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/calib}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
"$ROOT/himg_amd/bin/hbm_calib" 4 > "$ROOT/$OUT/hbm_calib.json" 2> "$ROOT/$OUT/hbm_calib.err"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/$c" -- \
      "$ROOT/himg_amd/bin/hbm_calib" 2 > "$ROOT/$OUT/$c.log" 2>&1
  echo "$c exit=$?"
done
cd "$ROOT"
python3 tools/summarize_calib.py "$OUT" > "$OUT/calibration.json" && cat "$OUT/calibration.json"
