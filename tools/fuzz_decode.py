#!/usr/bin/env python3
"""Decoder fuzz, long form (GPU box): N mutated streams per geometry and per form of the FRES
row index kernel (HIMG_OPT_COUNT_WAVE 0 / 1), GPU vs oracle: same accept / reject, same
pixels.  tests/test_gpu_fuzz.py is the bounded version that runs with the suite; this one's
output is kept as profiles/r04_fuzz_decode.txt.
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
engines = []
for cw in (0, 1):
    e = himg_amd.Engine(0)
    e.set_option("count_wave", cw)
    engines.append((cw, e))
rng = np.random.default_rng(7)
bad_cases = total = 0
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
        for cw, eng in engines:
            try:
                got = eng.decode(bad)
                ok = True
            except himg_amd.HimgError as e:
                ok, got = False, e.code
            total += 1
            if (rc == 0) != ok or (ok and not np.array_equal(got.ravel(), pix.ravel())):
                bad_cases += 1
                print("MISMATCH", kind, w, h, q, "mutation", t, "count_wave", cw, "oracle rc", rc, "gpu", "ok" if ok else got)
        acc += rc == 0; rej += rc != 0
    print(kind, w, h, q, "accepted", acc, "rejected", rej, flush=True)
print("decodes compared:", total, "mismatches:", bad_cases)
