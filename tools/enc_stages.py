#!/usr/bin/env python3
"""Stage times (HIP events per kernel) of the batch encode, no result check: for A/B runs of
environment knobs (HIMG_ROW_TOKENS, HIMG_EMIT_TOK_ROWS ...) on the GPU box.
args: width height batch [iters] [quality]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else w
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
it = int(sys.argv[4]) if len(sys.argv) > 4 else 10
q = int(sys.argv[5]) if len(sys.argv) > 5 else 50
eng = himg_amd.Engine(0)
frames = np.stack([himg_amd.synth("randtile", s, w, h) for s in range(B)])
d_frames = torch.from_numpy(frames).cuda()
cap = himg_amd.max_packed_size(w, h, 4)
d_out = torch.empty((B, cap), dtype=torch.uint8, device="cuda")
d_sizes = torch.zeros(B, dtype=torch.int32, device="cuda")
d_st = torch.ones(B, dtype=torch.int32, device="cuda")
def enc():
    eng.encode_device(d_frames, B, w, h, 4, 4, q, True, d_out, cap, d_sizes, d_st)
for _ in range(3):
    enc()
torch.cuda.synchronize()
ts = []
for _ in range(it):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); enc(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = np.array(ts)
print("env " + " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("HIMG_")))
print("encode of %d frames %dx%d q%d: min %.3f mean %.3f ms (%.1f Gpx/s)" % (B, w, h, q, ts.min(), ts.mean(), B * w * h / ts.mean() / 1e6))
eng.profile(True)
eng.profile_reset()
for _ in range(it):
    enc()
torch.cuda.synchronize()
st = eng.profile_read()
print({k: round(v[0] / it, 4) if isinstance(v, (tuple, list)) else v for k, v in sorted(st.items(), key=lambda kv: -(kv[1][0] if isinstance(kv[1], (tuple, list)) else kv[1]))})
