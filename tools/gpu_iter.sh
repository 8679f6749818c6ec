#!/bin/bash
# One GPU-box iteration: parity tests, per-row decode cycle stamps, the bench line.
#   tools/gpu_iter.sh <tag> [notest]
TAG=${1:-iter}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "${2:-}" != "notest" ]; then
  python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
  tail -3 $OUT/pytest.txt
fi
python3 tools/dec_stats_batch.py 4096 4096 64 > $OUT/dec_stats.txt 2>&1
python3 bench.py --no-cpu-baseline --no-extras --no-rows > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import json
l=[x for x in open('$OUT/bench.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('value %.0f enc %.0f dec %.0f ms %.3f' % (d['value'], d['encode_mpx_s'], d['decode_mpx_s'], d['ms_per_step'])); print(d['stages_ms'])
else:
    print(open('$OUT/bench.err').read()[-2000:])
PY
grep -E "decode of|clk_transform|clk_workgroup|clk_write" $OUT/dec_stats.txt
