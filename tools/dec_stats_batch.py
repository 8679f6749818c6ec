#!/usr/bin/env python3
"""Per-row cycle stamps of k_dec_row_fused while the whole GPU is busy: decode a
batch of identical frames and read frame 0's stats (args: width height batch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else w
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = himg_amd.Engine(0)
img = himg_amd.synth("randtile", 0, w, h)
packed = eng.encode(img, 50)
cap = (len(packed) + 255) // 256 * 256
d_in = torch.zeros((B, cap), dtype=torch.uint8, device="cuda")
d_in[:, :len(packed)] = torch.from_numpy(np.asarray(packed)).cuda()
d_pix = torch.empty((B, h, w, 4), dtype=torch.uint8, device="cuda")
d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
sizes = np.full(B, len(packed), np.uint32)
for _ in range(3):
    eng.decode_device(d_in, cap, sizes, B, w, h, 4, d_pix, d_st, 0)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    eng.decode_device(d_in, cap, sizes, B, w, h, 4, d_pix, d_st, 0)
ev1.record()
torch.cuda.synchronize()
print("decode of %d frames: %.3f ms" % (B, ev0.elapsed_time(ev1) / 5))
if not os.environ.get("HIMG_TIMING_BUILD"):   # (timing builds of experiments decode wrongly on purpose)
    assert not d_st.cpu().numpy().any(), "decode status"
rows = (h + 7) // 8
st = eng.debug_read("dec_stats", 0, (rows + 1) * 32, np.uint32, decoder=True).reshape(rows + 1, 8)
names = ["chunks", "rounds", "clk_transform/16 (slowest wave)", "clk_workgroup/16", "clk_sync/16", "clk_write/16", "pay_len", "out_size"]
fr = st[1:].astype(np.float64)
for i, n in enumerate(names):
    print("FRES %-32s mean %.1f min %.0f max %.0f" % (n, fr[:, i].mean(), fr[:, i].min(), fr[:, i].max()))
rc = eng.debug_read("rowcount_stats", 0, rows * 32, np.uint32, decoder=True).reshape(rows, 8).astype(np.float64) * 16
print("k_row_count cycles per row (slowest wave, frame 0): tables/staging %.0f, lead-in %.0f, to end of round 1 %.0f, fixpoint %.0f, row %.0f" % tuple(rc[:, :5].mean(axis=0)))
for q in range(4):
    print("  rows with r %% 4 == %d: tables/staging %.0f, lead-in %.0f, round 1 end %.0f, fixpoint end %.0f, row %.0f" % ((q,) + tuple(rc[q::4, :5].mean(axis=0))))
