#!/bin/bash
# Knock-out timing builds of k_dec_row_fused: the kernel with one class of operations
# removed (-DHIMG_DEC_KO=<bits>: 1 both ds_or of the write pass, 2 the second one only, 4 the transform's
# gather without the bank shared by a pair's two half-waves, 16 no pixel stores), timed per row by the
# kernel's own cycle stamps (tools/dec_stats_batch.py) under a full GPU.  They decode wrongly on purpose.
#   tools/dec_knockout.sh <outfile>      (GPU box; rebuilds the snapshot's library per variant)
OUT=${1:-gpurun_out/dec_knockout.txt}
: > "$OUT"
for ko in ${KOS:-0 1 2 4 16 17}; do
  echo "== HIMG_DEC_KO=$ko" >> "$OUT"
  if [ "$ko" = 0 ]; then unset HIMG_EXTRA_HIPCC_FLAGS; else export HIMG_EXTRA_HIPCC_FLAGS=-DHIMG_DEC_KO=$ko; fi
  python3 -c "from himg_amd import build; build.build_lib()" > /dev/null 2>&1
  HIMG_TIMING_BUILD=1 python3 tools/dec_stats_batch.py 4096 4096 64 2>/dev/null | grep -E "decode of|clk_transform|clk_workgroup|clk_write" >> "$OUT"
done
unset HIMG_EXTRA_HIPCC_FLAGS
python3 -c "from himg_amd import build; build.build_lib()" > /dev/null 2>&1
cat "$OUT"
