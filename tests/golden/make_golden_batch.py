#!/usr/bin/env python3
"""Golden tables for the BATCH workloads, from the REAL reference (oracle/_ref).

  tests/golden/batch_1920x1080_q50.json   BASELINE config 3: randtile seeds 0..255
  tests/golden/batch_4096x4096_q50.json   bench.py's frames (rank r uses seeds
                                           r*32 .. r*32+31, 8 ranks -> seeds 0..255)

Each row: [packed_size, stream FNV-1a-64, decoded FNV-1a-64].  Run in the build
container only (needs /root/reference compiled by `make -C oracle`); the tables are
data -- inputs come from the seeded generators of SURVEY.md Appendix C.1, expected
outputs from the reference itself (fresh Encoder per frame, trap T4).
Usage: python tests/golden/make_golden_batch.py [--workers N]
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(args):
    w, h, q, seed = args
    import himg_amd
    import oracle_lib as ol
    img = himg_amd.synth("randtile", seed, w, h)
    packed = ol.ref_encode(img, q, True)
    rc, dec = ol.ref_decode(packed, 1)
    assert rc == 0
    return seed, [int(packed.size), himg_amd.fnv1a64(packed), himg_amd.fnv1a64(dec)]


def table(w, h, q, n, workers):
    rows = [None] * n
    with ProcessPoolExecutor(workers) as pool:
        for seed, rec in pool.map(one, [(w, h, q, s) for s in range(n)]):
            rows[seed] = rec
            print(w, h, seed, rec, flush=True)
    out = {"kind": "randtile", "width": w, "height": h, "quality": q, "ycbcr": 1, "channels": 4,
           "columns": ["packed_size", "stream_fnv", "decoded_fnv"], "seeds": rows}
    json.dump(out, open(os.path.join(HERE, "batch_%dx%d_q%d.json" % (w, h, q)), "w"), indent=0)


def main():
    import oracle_lib as ol
    assert ol.have_ref(), "build oracle/_ref first: make -C oracle"
    workers = 6
    if "--workers" in sys.argv:
        workers = int(sys.argv[sys.argv.index("--workers") + 1])
    table(1920, 1080, 50, 256, workers)
    table(4096, 4096, 50, 256, workers)


if __name__ == "__main__":
    main()
