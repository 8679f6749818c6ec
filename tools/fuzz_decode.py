#!/usr/bin/env python3
"""One-off decoder fuzz (GPU box): N single-bit flips per stream, GPU vs oracle.
Usage: python tools/fuzz_decode.py [N]"""
import os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import himg_amd
import oracle_lib as ol

def chunks(stream):
    b, out, i = bytes(stream), {}, 12
    while i + 8 <= len(b):
        sz = struct.unpack("<I", b[i + 4:i + 8])[0]
        out[b[i:i + 4].decode()] = (i + 8, sz)
        i += 8 + sz
    return out

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = himg_amd.Engine(0)
rng = np.random.default_rng(7)
bad_cases = 0
for kind, w, h, q in [("randtile", 4096, 64, 50), ("randtile", 4096, 32, 90), ("gradn", 1024, 256, 70),
                      ("rand", 512, 128, 50), ("randtile", 200, 116, 70), ("randtile", 4400, 40, 90),
                      ("rand", 4400, 16, 90), ("randtile", 8192, 64, 70), ("rand", 16384, 40, 90)]:
    img = himg_amd.synth(kind, 5, w, h)
    good = ol.oracle_encode(img, q, True)
    if ol.oracle_decode(good)[0] != 0:
        print("skip (T2)", kind, w, h, q); continue
    ch = chunks(good)
    acc = rej = 0
    for t in range(N):
        bad = good.copy()
        # Mostly the entropy-coded payloads; every 7th mutation hits the companding
        # map or the shift table instead (large dequantised coefficients: the
        # transform's int32 path and its int16 wrap).
        off, sz = ch["FRES" if t % 3 else "LRES"]
        if t % 7 == 3:
            off, sz = ch["FMAP" if t % 2 else "QCFG"]
        nflip = 1 + (t % 5 == 0)
        # every fourth mutation goes for the serialised Huffman tree at the head of the chunk
        span = min(sz, 340) if (t % 4 == 1 and t % 7 != 3) else sz
        for _ in range(nflip):
            bad[int(rng.integers(off, off + span))] ^= 1 << int(rng.integers(0, 8))
        rc, pix = ol.oracle_decode(bad)
        try:
            got = eng.decode(bad)
            ok = True
        except himg_amd.HimgError as e:
            ok, got = False, e.code
        if (rc == 0) != ok or (ok and not np.array_equal(got.ravel(), pix.ravel())):
            bad_cases += 1
            print("MISMATCH", kind, w, h, q, "mutation", t, "oracle rc", rc, "gpu", "ok" if ok else got)
        acc += rc == 0; rej += rc != 0
    print(kind, w, h, q, "accepted", acc, "rejected", rej)
print("mismatches:", bad_cases)
