#!/usr/bin/env python3
"""Single-frame latency through the device-resident API, and the per-kernel times of
one frame (args: width height) (GPU box)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
eng = himg_amd.Engine(0)
img = himg_amd.synth("randtile", 0, W, H)
d = torch.from_numpy(img).to("cuda:0")[None]
cap = himg_amd.max_packed_size(W, H, 4)
d_out = torch.empty((1, cap), dtype=torch.uint8, device="cuda:0")
d_sz = torch.zeros(1, dtype=torch.int32, device="cuda:0")
d_st = torch.zeros(1, dtype=torch.int32, device="cuda:0")
d_pix = torch.empty((1, H, W, 4), dtype=torch.uint8, device="cuda:0")
enc = lambda: eng.encode_device(d, 1, W, H, 4, 4, 50, True, d_out, cap, d_sz, d_st, 0)
enc(); torch.cuda.synchronize()
hs = d_sz.cpu().numpy().astype(np.uint32)
dec = lambda: eng.decode_device(d_out, cap, hs, 1, W, H, 4, d_pix, d_st, 0)
dec(); torch.cuda.synchronize()
res = {}
for name, fn in (("encode", enc), ("decode", dec)):
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    eng.profile_reset(); eng.profile(True)
    for _ in range(5):
        fn(); torch.cuda.synchronize()
    eng.profile(False)
    # per call (5 profiled calls): a kernel launched twice per call counts with both launches
    st = {k: round(v[0] / 5 * 1e3, 1) for k, v in sorted(eng.profile_read().items(), key=lambda kv: -kv[1][0])}
    res[name] = {"latency_ms": {"min": round(min(ts), 3), "mean": round(sum(ts) / len(ts), 3)}, "kernels_us": st}
print(json.dumps(res, indent=1))
