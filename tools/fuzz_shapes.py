#!/usr/bin/env python3
"""One-off parity sweep (GPU box): random shapes / channel counts / strides /
qualities / colour modes, GPU encode + decode vs the oracle.
Usage: python tools/fuzz_shapes.py [N] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import himg_amd
import oracle_lib as ol

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
eng = himg_amd.Engine(0)
bad = dec_ok = dec_rej = 0
for t in range(N):
    big = t % 10 == 0
    w = int(rng.integers(1, 2600 if big else 420))
    h = int(rng.integers(1, 300 if big else 260))
    ch = int(rng.integers(1, 5))
    stride = ch + int(rng.integers(0, 3)) * (t % 4 == 0)
    q = int(rng.choice([0, 5, 10, 30, 50, 70, 90, 100]))
    ycbcr = bool(rng.integers(0, 2))
    kind = ["randtile", "gradn", "rand", "grad"][int(rng.integers(0, 4))]
    base = himg_amd.synth(kind, int(rng.integers(0, 1000)), max(w, 8), max(h, 8))[:h, :w]
    img = np.zeros((h, w, stride), np.uint8)
    img[:, :, :min(stride, 4)] = base[:, :, :min(stride, 4)]
    if stride > 4:
        img[:, :, 4:] = rng.integers(0, 256, (h, w, stride - 4), dtype=np.uint8)
    img = np.ascontiguousarray(img)
    want = ol.oracle_encode(img, q, ycbcr, channels=ch, stride=stride)
    tag = "%dx%d ch%d stride%d q%d ycbcr%d %s" % (w, h, ch, stride, q, ycbcr, kind)
    try:
        got = eng.encode(img, q, ycbcr, channels=ch, pixel_stride=stride)
    except himg_amd.HimgError as e:
        print("ENCODE ERROR", tag, e); bad += 1; continue
    if not np.array_equal(got, want):
        print("STREAM MISMATCH", tag, got.size, want.size); bad += 1; continue
    rc, pix = ol.oracle_decode(want)
    try:
        gp = eng.decode(want); ok = True
    except himg_amd.HimgError as e:
        ok = False
    if w % 8 == 0 or rc != 0:   # W % 8 != 0: the reference's own decode is undefined (trap T9)
        if (rc == 0) != ok or (ok and not np.array_equal(gp.ravel(), pix.ravel())):
            print("DECODE MISMATCH", tag, "oracle rc", rc, "gpu ok", ok); bad += 1
    dec_ok += rc == 0; dec_rej += rc != 0
print("cases", N, "decodable", dec_ok, "rejected", dec_rej, "mismatches", bad)
