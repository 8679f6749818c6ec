#!/usr/bin/env python3
"""Encode / decode time of one launch over B distinct frames (randtile seeds 0..B-1),
not profiled, for A/B runs of environment knobs (GPU box).  args: width height batch [iters]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else w
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
it = int(sys.argv[4]) if len(sys.argv) > 4 else 20
eng = himg_amd.Engine(0)
frames = np.stack([himg_amd.synth("randtile", s, w, h) for s in range(B)])
d_frames = torch.from_numpy(frames).cuda()
cap = himg_amd.max_packed_size(w, h, 4)
d_out = torch.empty((B, cap), dtype=torch.uint8, device="cuda")
d_sizes = torch.zeros(B, dtype=torch.int32, device="cuda")
d_st = torch.ones(B, dtype=torch.int32, device="cuda")
d_pix = torch.empty((B, h, w, 4), dtype=torch.uint8, device="cuda")
def enc():
    eng.encode_device(d_frames, B, w, h, 4, 4, 50, True, d_out, cap, d_sizes, d_st)
enc(); torch.cuda.synchronize()
sizes = d_sizes.cpu().numpy().astype(np.uint32)
def dec():
    eng.decode_device(d_out, cap, sizes, B, w, h, 4, d_pix, d_st, 0)
res = {}
for name, fn in (("encode", enc), ("decode", dec)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = np.array(ts)
    res[name] = ts
    print("%s of %d frames %dx%d: min %.3f mean %.3f max %.3f ms  (%.1f Gpx/s at the mean)" % (
        name, B, w, h, ts.min(), ts.mean(), ts.max(), B * w * h / ts.mean() / 1e6))
assert not d_st.cpu().numpy().any()
# What was timed is the real thing: frame 0's stream and pixels against the golden table
# (tests/golden/batch_<w>x<h>_q50.json, made from the real reference) where there is one.
import json
gpath = os.path.join(ROOT, "tests", "golden", "batch_%dx%d_q50.json" % (w, h))
if os.path.exists(gpath):
    size0, stream_fnv, decoded_fnv = json.load(open(gpath))["seeds"][0]   # seed 0 = frame 0
    stream = d_out[0, : int(sizes[0])].cpu().numpy()
    assert int(sizes[0]) == size0 and himg_amd.fnv1a64(stream) == stream_fnv, "stream differs from the reference"
    assert himg_amd.fnv1a64(d_pix[0].cpu().numpy()) == decoded_fnv, "pixels differ from the reference"
    print("frame 0 checked against the golden table")
else:
    print("no golden table for %dx%d: results not checked" % (w, h))
