#!/usr/bin/env python3
"""One large frame through the row-sharded encode / decode on one GPU, a few times, with
nothing else going on -- for rocprofv3 --kernel-trace (tools/trace_timeline.py reads it).
  rows_loop.py decode|encode [width height repetitions]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
from himg_amd import sharded
what = sys.argv[1] if len(sys.argv) > 1 else "decode"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
H = int(sys.argv[3]) if len(sys.argv) > 3 else W
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
img = himg_amd.synth("randtile", 0, W, H)
eng = himg_amd.Engine(0)
d = torch.from_numpy(img).to("cuda:0")
back = sharded.EngineBackend(eng, d, 0, W, H, 50, True)
out = sharded.encode_sharded(back, H // 8, W // 8, 4, True, host=False)
for _ in range(n):
    if what == "encode":
        sharded.encode_sharded(back, H // 8, W // 8, 4, True, host=False)
    else:
        ok, _ = sharded.decode_sharded(eng, out, W, H, 4, gather=False, device="cuda:0")
        assert ok
    torch.cuda.synchronize()
