"""Shared helpers for the golden-vector tests."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(HERE, "golden")

with open(os.path.join(GOLDEN_DIR, "golden.json")) as _f:
    GOLDEN = json.load(_f)


def cases(max_pixels=None, min_pixels=0, decodable=None):
    out = []
    for name, r in sorted(GOLDEN.items()):
        px = r["width"] * r["height"]
        if max_pixels is not None and px > max_pixels:
            continue
        if px < min_pixels:
            continue
        if decodable is not None and r["decodes"] != decodable:
            continue
        out.append(name)
    return out


def make_input(rec):
    import himg_amd
    img = himg_amd.synth(rec["kind"], rec["seed"], rec["width"], rec["height"])
    if rec["channels"] != 4:
        img = np.ascontiguousarray(img[:, :, :rec["channels"]])
    return img


def fixture(rec):
    with open(os.path.join(GOLDEN_DIR, rec["fixture"]), "rb") as f:
        return np.frombuffer(f.read(), np.uint8).copy()
