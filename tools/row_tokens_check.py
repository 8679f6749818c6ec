"""FRES rows through the token stream (k_tok + k_emit_tok) against the dense-plane kernels and
the oracle: final bytes, the slots expanded into symbols again, the histogram.  Runs on the GPU box."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import himg_amd  # noqa: E402

try:
    import oracle_lib as ol
except Exception:  # pragma: no cover
    ol = None


def images(w, h):
    yield "randtile", himg_amd.synth("randtile", 1, w, h)
    yield "grad", himg_amd.synth("grad", 2, w, h)
    yield "rand", himg_amd.synth("rand", 3, w, h)
    z = np.zeros((h, w, 4), np.uint8)
    yield "zero", z
    s = z.copy()
    s[h // 2:, :, :] = himg_amd.synth("rand", 4, w, h)[h // 2:, :, :]
    s[: h // 2, w - 9, 1] = 200   # a lone column of detail: very long runs
    yield "sparse", s


def main():
    eng = himg_amd.Engine(0)
    bad = 0
    sizes = [(512, 64), (512, 512), (1024, 256), (2048, 128), (4096, 64), (4096, 512), (1920, 136), (200, 72), (520, 40)]
    if "--big" in sys.argv:
        sizes += [(4096, 4096)]
    for (w, h) in sizes:
        for name, img in images(w, h):
            for q in (10, 50, 90, 100):
                for ycc in (True, False):
                    eng.set_option("row_tokens", 0)
                    a = eng.encode(img, q, ycc)
                    eng.set_option("row_tokens", 1)
                    if "-v" in sys.argv:
                        print("encode", w, h, name, q, ycc, flush=True)
                    b = eng.encode(img, q, ycc)
                    ok = a.size == b.size and np.array_equal(a, b)
                    if q in (50, 100):
                        eng.set_option("row_tokens", 2)
                        b2 = eng.encode(img, q, ycc)
                        ok = ok and b2.size == a.size and np.array_equal(a, b2)
                    sym_ok = True
                    if ok and ol is not None and w * h <= 512 * 512 and q == 50:
                        want, tr = ol.oracle_encode(img, q, ycc, trace=True)
                        got = eng.debug_read("fres_tok_sym", 0, tr["fres_sym"].size)
                        sym_ok = np.array_equal(got, tr["fres_sym"]) and np.array_equal(b, want)
                        if not np.array_equal(eng.debug_read("fres_hist", 0, 261 * 4, np.uint32), tr["fres_hist"]):
                            sym_ok = False
                    if not (ok and sym_ok):
                        bad += 1
                        first = int(np.argmax(a[: min(a.size, b.size)] != b[: min(a.size, b.size)])) if a.size and b.size else -1
                        print("MISMATCH %dx%d %s q%d ycc%d sizes %d/%d first diff %d sym_ok %s" % (w, h, name, q, ycc, a.size, b.size, first, sym_ok))
            print("%dx%d %s done" % (w, h, name), flush=True)
    print("row_tokens_check: %d mismatches" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
