#!/bin/bash
# Collect rocprofv3 PMC counters for the bench workload, one counter group per
# run (MI355X_MICROARCH.md: 8 SQ slots, FETCH_SIZE and WRITE_SIZE in separate
# passes; never combined with trace domains other than kernel-trace).
# Usage (on the GPU box, from the repo root): tools/profile_pmc.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} \
      > "$ROOT/$OUT/$name.log" 2>&1
  echo "$name exit=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS
# Dynamic evidence for the issue model: thread-cycles spent in VALU instructions (per
# executed instruction and active lane), the integer instruction classes, LDS traffic by kind.
run sq3 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_CYCLES_SALU
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
python3 "$ROOT/tools/summarize_pmc.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt" 2>&1
tail -n 60 "$ROOT/$OUT/summary.txt"
