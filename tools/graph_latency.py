#!/usr/bin/env python3
"""Single-frame latency with the launch chain captured in a HIP graph.

The encode of one 4096x4096 frame is eleven dependent launches on two streams, the decode a
dozen on three; this captures either chain once (torch.cuda.graph -> hipStreamBeginCapture on
torch's current stream; the engine's side streams join the capture through the fork / join
events the chain already has) and replays it, next to the same calls issued one by one.
Prints one JSON object; a capture the runtime refuses is reported as such, not hidden."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import himg_amd  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
Q = 50
dev = torch.device("cuda", 0)
eng = himg_amd.Engine(0)
img = himg_amd.synth("randtile", 0, W, H)
d_frame = torch.from_numpy(img).to(dev).reshape(1, H, W, 4)
cap = himg_amd.max_packed_size(W, H, 4)
d_out = torch.zeros((1, cap), dtype=torch.uint8, device=dev)
d_sizes = torch.zeros(1, dtype=torch.int32, device=dev)
d_se = torch.zeros(1, dtype=torch.int32, device=dev)
d_sd = torch.zeros(1, dtype=torch.int32, device=dev)
d_pix = torch.zeros((1, H, W, 4), dtype=torch.uint8, device=dev)


def enc(stream):
    eng.encode_device(d_frame, 1, W, H, 4, 4, Q, True, d_out, cap, d_sizes, d_se, stream)


enc(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
h_sizes = d_sizes.cpu().numpy().astype(np.uint32)
ref_stream = d_out[0, : int(h_sizes[0])].cpu().numpy().copy()


def dec(stream):
    eng.decode_device(d_out, cap, h_sizes, 1, W, H, 4, d_pix, d_sd, stream)


dec(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ref_pix = d_pix.cpu().numpy().copy()


def warm():
    """Clocks up: a second of back-to-back work before anything is timed (an idle GPU answers the
    first dozen launches at its idle clock -- 0.55 instead of 0.39 ms for the same encode)."""
    t = time.perf_counter()
    while time.perf_counter() - t < 1.0:
        for _ in range(20):
            enc(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()


def lat(fn, reps=50):
    warm()
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return {"min": round(ts[0], 4), "median": round(ts[len(ts) // 2], 4), "mean": round(sum(ts) / len(ts), 4)}


out = {"width": W, "height": H, "quality": Q}
for name, fn, check in (("encode", enc, lambda: np.array_equal(d_out[0, : ref_stream.size].cpu().numpy(), ref_stream)),
                        ("decode", dec, lambda: np.array_equal(d_pix.cpu().numpy(), ref_pix))):
    res = {"launches_one_by_one_ms": lat(lambda: fn(torch.cuda.current_stream().cuda_stream))}
    try:
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            fn(s.cuda_stream)                     # warm-up on the capture stream (workspace, attributes)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
                fn(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if name == "encode":
            d_out.zero_()
        else:
            d_pix.zero_()
        g.replay()
        torch.cuda.synchronize()
        res["graph_bit_exact"] = bool(check())
        res["graph_replay_ms"] = lat(g.replay)
    except Exception as e:   # noqa: BLE001 -- the point is to report what the runtime says
        res["graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    out[name] = res
print(json.dumps(out, indent=1))
