#!/usr/bin/env python3
"""LRES chain statistics of every frame of a batch of DISTINCT frames (randtile seeds
0..B-1): chunks settled by the cheap test vs warm restarts, rounds, cycles (GPU box).
args: width height batch"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else w
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
eng = himg_amd.Engine(0)
frames = np.stack([himg_amd.synth("randtile", s, w, h) for s in range(B)])
d_frames = torch.from_numpy(frames).cuda()
cap = himg_amd.max_packed_size(w, h, 4)
d_out = torch.empty((B, cap), dtype=torch.uint8, device="cuda")
d_sizes = torch.zeros(B, dtype=torch.int32, device="cuda")
d_st = torch.ones(B, dtype=torch.int32, device="cuda")
eng.encode_device(d_frames, B, w, h, 4, 4, 50, True, d_out, cap, d_sizes, d_st)
torch.cuda.synchronize()
sizes = d_sizes.cpu().numpy().astype(np.uint32)
d_pix = torch.empty((B, h, w, 4), dtype=torch.uint8, device="cuda")
eng.decode_device(d_out, cap, sizes, B, w, h, 4, d_pix, d_st, 0)
torch.cuda.synchronize()
assert not d_st.cpu().numpy().any()
rows = (h + 7) // 8
for f in range(B):
    st = eng.debug_read("dec_stats", f, (rows + 1) * 32, np.uint32, decoder=True).reshape(rows + 1, 8)[0]
    print("frame %2d: spec chunks %d rounds %d (max %d, %d cyc) | settled+restarted %d, restart rounds %d (max %d, %d cyc)" % (
        f, st[0], st[1], st[4], int(st[6]) * 16, st[2], st[3], st[5], int(st[7]) * 16))
