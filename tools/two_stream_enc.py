#!/usr/bin/env python3
"""Experiment: one batch encode of B frames against two half batches on two streams (two contexts),
optionally staggered -- do kernels of different kinds (HBM-bound pixel stage, issue-bound tokeniser)
overlap when the halves run side by side?  args: width height batch"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else w
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
frames = np.stack([himg_amd.synth("randtile", s, w, h) for s in range(B)])
d_frames = torch.from_numpy(frames).cuda()
cap = himg_amd.max_packed_size(w, h, 4)
d_out = torch.empty((B, cap), dtype=torch.uint8, device="cuda")
d_sizes = torch.zeros(B, dtype=torch.int32, device="cuda")
d_st = torch.ones(B, dtype=torch.int32, device="cuda")
e0, e1, e2 = himg_amd.Engine(0), himg_amd.Engine(0), himg_amd.Engine(0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one():
    e0.encode_device(d_frames, B, w, h, 4, 4, 50, True, d_out, cap, d_sizes, d_st)

def two(split):
    n1 = split
    e1.encode_device(d_frames[:n1], n1, w, h, 4, 4, 50, True, d_out[:n1], cap, d_sizes[:n1], d_st[:n1], stream=s1.cuda_stream)
    e2.encode_device(d_frames[n1:], B - n1, w, h, 4, 4, 50, True, d_out[n1:], cap, d_sizes[n1:], d_st[n1:], stream=s2.cuda_stream)

def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        s1.synchronize(); s2.synchronize()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sum(ts) / len(ts)

print("one batch of %d: min %.3f mean %.3f ms" % ((B,) + timeit(one)))
for split in (B // 2, B // 4, 3 * B // 8):
    cur = torch.cuda.current_stream()
    def fn():
        s1.wait_stream(cur); s2.wait_stream(cur)
        two(split)
    print("two streams %d + %d: min %.3f mean %.3f ms" % ((split, B - split) + timeit(fn)))
