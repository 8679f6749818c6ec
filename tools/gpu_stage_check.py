#!/usr/bin/env python3
"""Stage-by-stage parity report of the HIP path against the CPU oracle.

Development aid for the GPU box: for a list of synthetic cases it prints, for
every intermediate product (box averages, low-res plane, LRES/FRES symbols,
histograms, code lengths, row sizes, final stream, decoder symbols, pixels),
whether the GPU result equals the oracle's, and where the first mismatch is.
Usage: python tools/gpu_stage_check.py [--cases small|all]
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import himg_amd  # noqa: E402
import oracle_lib as ol  # noqa: E402


def first_diff(a, b):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    if a.size != b.size:
        return "size %d vs %d" % (a.size, b.size)
    d = np.nonzero(a != b)[0]
    if d.size == 0:
        return None
    i = int(d[0])
    return "%d mismatches, first at %d: gpu=%s oracle=%s" % (d.size, i, a[i], b[i])


def check(name, gpu, orc, fails):
    d = first_diff(gpu, orc)
    print("    %-16s %s" % (name, "ok" if d is None else "MISMATCH " + d))
    if d is not None:
        fails.append(name)


def run_case(eng, kind, seed, w, h, q, ycbcr=True):
    print("case %s seed=%d %dx%d q=%d ycbcr=%d" % (kind, seed, w, h, q, ycbcr))
    fails = []
    img = himg_amd.synth(kind, seed, w, h)
    packed_o, tr = ol.oracle_encode(img, q, ycbcr, trace=True)
    t0 = time.time()
    try:
        packed_g = eng.encode(img, q, ycbcr)
    except himg_amd.HimgError as e:
        print("    encode raised", e)
        packed_g = None
        fails.append("encode")
    t1 = time.time()
    C, rows, cols = 4, tr["rows"], tr["cols"]
    n_plane = C * rows * cols
    check("avg", eng.debug_read("avg", 0, n_plane), tr["avg"], fails)
    check("lowres", eng.debug_read("lowres", 0, n_plane), tr["lowres"], fails)
    check("lres_sym", eng.debug_read("lres_sym", 0, tr["lres_sym"].size), tr["lres_sym"], fails)
    check("fres_sym", eng.debug_read("fres_sym", 0, tr["fres_sym"].size), tr["fres_sym"], fails)
    for k in ("lres_hist", "fres_hist", "lres_len", "fres_len"):
        check(k, eng.debug_read(k, 0, 261 * 4, np.uint32), tr[k], fails)
    for k in ("lres_code", "fres_code"):
        check(k, eng.debug_read(k, 0, 261 * 8, np.uint64), tr[k], fails)
    check("fres_row_bytes", eng.debug_read("fres_row_bytes", 0, rows * 4, np.uint32),
          tr["fres_row_bytes"], fails)
    if packed_g is not None:
        check("stream", packed_g, packed_o, fails)
        print("    encode wall %.1f ms, %d bytes, fnv %s" % ((t1 - t0) * 1e3, packed_g.size,
                                                            himg_amd.fnv1a64(packed_g)))
    # Decode the ORACLE's stream so decoder problems are isolated from encoder ones.
    rc, dt = ol.oracle_decode_trace(packed_o)
    try:
        pix = eng.decode(packed_o)
        if rc != 0:
            print("    decode: GPU accepted a stream the oracle rejects (rc=%d)" % rc)
            fails.append("decode-accept")
        else:
            check("dec lres_sym", eng.debug_read("lres_sym", 0, dt["lres_sym"].size, decoder=True),
                  dt["lres_sym"], fails)
            check("dec lowres", eng.debug_read("lowres", 0, n_plane, decoder=True), dt["lowres"], fails)
            check("dec fres_sym", eng.debug_read("fres_sym", 0, dt["fres_sym"].size, decoder=True),
                  dt["fres_sym"], fails)
            check("pixels", pix, dt["pixels"], fails)
    except himg_amd.HimgError as e:
        if rc != 0 and e.code == himg_amd.HIMG_ERR_FORMAT:
            print("    decode: rejected like the reference (oracle rc=%d)" % rc)
        else:
            print("    decode raised", e, "oracle rc", rc)
            fails.append("decode")
            if rc == 0:
                for k, key in (("dec lres_sym", "lres_sym"), ("dec lowres", "lowres"), ("dec fres_sym", "fres_sym")):
                    try:
                        n = dt[key].size
                        check(k, eng.debug_read(key, 0, n, decoder=True), dt[key], fails)
                    except Exception as e2:  # noqa: BLE001
                        print("    ", k, "unreadable", e2)
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="small")
    args = ap.parse_args()
    cases = [("gradn", 0, 64, 64, 50), ("rand", 0, 64, 64, 50), ("randtile", 0, 64, 64, 50),
             ("grad", 0, 64, 64, 50), ("randtile", 3, 200, 120, 50), ("gradn", 0, 512, 512, 50),
             ("randtile", 1, 512, 256, 90), ("randtile", 2, 256, 512, 10)]
    if args.cases == "all":
        cases += [("randtile", 0, 1920, 1080, 50), ("randtile", 0, 4096, 4096, 50),
                  ("rand", 0, 2048, 2048, 50), ("grad", 0, 512, 512, 50)]
    eng = himg_amd.Engine(0)
    bad = {}
    for c in cases:
        try:
            f = run_case(eng, *c)
        except Exception:  # noqa: BLE001
            traceback.print_exc()
            f = ["exception"]
        if f:
            bad[str(c)] = f
    print("SUMMARY:", "all stages match" if not bad else bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
