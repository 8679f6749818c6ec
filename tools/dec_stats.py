#!/usr/bin/env python3
"""Print k_dec_huff diagnostics for one frame (args: width [height [save.npy]]) (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import himg_amd
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else w
eng = himg_amd.Engine(0)
img = himg_amd.synth("randtile", 0, w, h)
packed = eng.encode(img, 50)
eng.decode(packed)
rows = (h + 7) // 8
st = eng.debug_read("dec_stats", 0, (rows + 1) * 32, np.uint32, decoder=True).reshape(rows + 1, 8)
names = ["chunks", "rounds", "clk_transform/16, slowest wave (lres: fix chunks)", "clk_workgroup/16 (lres: fix rounds; unfused: clk_round1)", "clk_sync/16", "clk_write/16", "pay_len", "out_size"]
print("LRES: chunks %d, fixpoint rounds %d (slowest chunk %d); corrected chunks %d, rounds %d (slowest %d)" % (st[0][0], st[0][1], st[0][4], st[0][2], st[0][3], st[0][5]))
print("LRES: longest fixpoint %d cycles (speculative), %d cycles (corrected)" % (int(st[0][6]) * 16, int(st[0][7]) * 16))
ps = eng.debug_read("parse_stats", 0, 16, np.uint32, decoder=True)
print("k_dec_parse cycles: serial %d, lut %d, sub %d, grp %d" % tuple((ps.astype(np.int64) * 16).tolist()))
rc = eng.debug_read("rowcount_stats", 0, rows * 32, np.uint32, decoder=True).reshape(rows, 8).astype(np.float64) * 16
print("k_row_count cycles (slowest wave): tables+staging %.0f, lead-in %.0f, to end of round 1 %.0f, fixpoint %.0f, workgroup %.0f" % tuple(rc[:, :5].mean(axis=0)))
rr = st[1:, 1].astype(np.int64)
print("FRES per row: rounds mean %.2f, lanes re-joined in rounds>=2 mean %.1f, lanes fully re-decoded mean %.1f max %d" % (
    (rr & 255).mean(), ((rr >> 8) & 4095).mean(), ((rr >> 20) & 4095).mean(), ((rr >> 20) & 4095).max()))
fr = st[1:].astype(np.float64)
for i, n in enumerate(names):
    print("FRES %-14s mean %.1f min %.0f max %.0f" % (n, fr[:, i].mean(), fr[:, i].min(), fr[:, i].max()))
if len(sys.argv) > 3:
    np.save(sys.argv[3], st)
    r = st[1:, 1]
    print("rounds histogram:", np.bincount(np.minimum(r, 60) // 5))
    print("rows with most rounds:", np.argsort(-r.astype(int))[:16].tolist(), np.sort(r)[::-1][:16].tolist())
