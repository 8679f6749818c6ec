#!/usr/bin/env python3
"""Regenerate the golden vectors from the REAL reference (oracle/_ref).

Run in the build container only (it needs /root/reference to have been compiled
by `make -C oracle`).  Outputs, all committed:
  tests/golden/golden.json      sizes / FNV-1a-64 hashes / PSNR per case
  tests/golden/<case>.himg      full streams of the 64x64 cases (a few KB each)
The fixtures are data (inputs are regenerated from the seeded generators of
SURVEY.md Appendix C.1; expected outputs come from the reference itself).
Usage: python tests/golden/make_golden.py [--big]   (--big adds 16384x16384)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import himg_amd  # noqa: E402
import oracle_lib as ol  # noqa: E402


def chunk_sizes(stream):
    """Walk the RIFF chunks (decoder.cpp:428-461) -> {tag: size}."""
    out, idx = {}, 12
    while idx + 8 <= len(stream):
        tag = bytes(stream[idx:idx + 4]).decode("latin1")
        sz = int.from_bytes(bytes(stream[idx + 4:idx + 8]), "little")
        out[tag] = sz
        idx += 8 + sz
    return out


def case(kind, seed, w, h, q, ycbcr=True, channels=4, save=False):
    img = himg_amd.synth(kind, seed, w, h)
    if channels != 4:
        img = np.ascontiguousarray(img[:, :, :channels])
    packed = ol.ref_encode(img, q, ycbcr)
    rc, dec = ol.ref_decode(packed, 0)
    name = "%s_s%d_%dx%d_q%d%s%s" % (kind, seed, w, h, q, "" if ycbcr else "_rgb",
                                     "" if channels == 4 else "_c%d" % channels)
    rec = {
        "kind": kind, "seed": seed, "width": w, "height": h, "quality": q,
        "ycbcr": int(ycbcr), "channels": channels,
        "input_fnv": himg_amd.fnv1a64(img), "packed_size": int(packed.size),
        "chunks": chunk_sizes(packed), "stream_fnv": himg_amd.fnv1a64(packed),
        "decodes": rc == 0,
    }
    if rc == 0:
        rec["decoded_fnv"] = himg_amd.fnv1a64(dec)
        rec["psnr"] = round(himg_amd.psnr(img, dec), 4)
    if save:
        with open(os.path.join(HERE, name + ".himg"), "wb") as f:
            f.write(packed.tobytes())
        rec["fixture"] = name + ".himg"
    print(name, rec["packed_size"], rec["stream_fnv"], rec.get("decoded_fnv"), rec.get("psnr"))
    return name, rec


def main():
    assert ol.have_ref(), "build oracle/_ref first: make -C oracle"
    big = "--big" in sys.argv
    table = {}
    for kind in ("grad", "gradn", "rand", "randtile"):
        n, r = case(kind, 0, 64, 64, 50, save=True)
        table[n] = r
    for q in (0, 10, 30, 70, 90, 100):
        n, r = case("randtile", 0, 64, 64, q, save=(q in (90, 100)))
        table[n] = r
    extra = [
        ("randtile", 5, 64, 64, 50, False, 4), ("randtile", 5, 64, 64, 50, True, 3),
        ("randtile", 5, 64, 64, 50, True, 1), ("gradn", 1, 136, 72, 50, True, 4),
        ("randtile", 7, 64, 8, 50, True, 4), ("randtile", 7, 8, 64, 90, True, 4),
        ("randtile", 9, 200, 116, 70, True, 4), ("gradn", 0, 512, 512, 50, True, 4),
        ("grad", 0, 512, 512, 50, True, 4), ("randtile", 0, 1920, 1080, 50, True, 4),
        ("randtile", 1, 1920, 1080, 50, True, 4), ("randtile", 255, 1920, 1080, 50, True, 4),
        ("rand", 0, 2048, 2048, 50, True, 4), ("rand", 0, 4096, 4096, 50, True, 4),
    ]
    for kind, seed, w, h, q, ycbcr, ch in extra:
        n, r = case(kind, seed, w, h, q, ycbcr, ch, save=(w * h <= 64 * 64 and ch != 4))
        table[n] = r
    for q in (10, 30, 50, 70, 90):
        n, r = case("randtile", 0, 4096, 4096, q)
        table[n] = r
    path = os.path.join(HERE, "golden.json")
    if big:
        n, r = case("randtile", 0, 16384, 16384, 50)
        table[n] = r
    elif os.path.exists(path):
        old = json.load(open(path))
        for k, v in old.items():
            if k not in table:
                table[k] = v  # keep earlier --big results
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
