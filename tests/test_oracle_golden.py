"""CPU: pin the oracle (our C restatement) to the reference.

* against the golden vectors recorded from the real reference
  (tests/golden/golden.json + *.himg, made by tests/golden/make_golden.py);
* against the real reference itself when oracle/_ref has been built here.
"""
import numpy as np
import pytest

import himg_amd
import oracle_lib as ol
from golden_util import GOLDEN, cases, fixture, make_input

SMALL = 1920 * 1080


@pytest.mark.parametrize("name", cases(max_pixels=SMALL))
def test_oracle_encode_matches_golden(name):
    rec = GOLDEN[name]
    img = make_input(rec)
    assert himg_amd.fnv1a64(img) == rec["input_fnv"]
    packed = ol.oracle_encode(img, rec["quality"], bool(rec["ycbcr"]))
    assert packed.size == rec["packed_size"]
    assert himg_amd.fnv1a64(packed) == rec["stream_fnv"]
    if "fixture" in rec:
        assert np.array_equal(packed, fixture(rec))


@pytest.mark.parametrize("name", cases(max_pixels=SMALL))
def test_oracle_decode_matches_golden(name):
    rec = GOLDEN[name]
    img = make_input(rec)
    packed = fixture(rec) if "fixture" in rec else ol.oracle_encode(img, rec["quality"], bool(rec["ycbcr"]))
    rc, dec = ol.oracle_decode(packed, threads=2)
    if not rec["decodes"]:
        assert rc != 0  # trap T2: the reference refuses its own stream
        return
    assert rc == 0
    assert himg_amd.fnv1a64(dec) == rec["decoded_fnv"]
    assert round(himg_amd.psnr(img, dec), 4) == pytest.approx(rec["psnr"], abs=1e-4)


def test_oracle_headline_config_4096():
    """BASELINE config 2: 4096x4096 randtile q50, byte-exact size and hashes."""
    rec = GOLDEN["randtile_s0_4096x4096_q50"]
    img = make_input(rec)
    packed = ol.oracle_encode(img, 50, True)
    assert packed.size == 17227700 == rec["packed_size"]
    assert himg_amd.fnv1a64(packed) == "65c2fb5345506268" == rec["stream_fnv"]
    rc, dec = ol.oracle_decode(packed, threads=0)
    assert rc == 0 and himg_amd.fnv1a64(dec) == "dd3685000a721519"


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("kind,seed,w,h,q,ycbcr,ch", [
    ("gradn", 3, 64, 64, 50, True, 4), ("rand", 4, 72, 40, 20, True, 4),
    ("randtile", 5, 136, 64, 60, True, 4), ("randtile", 6, 64, 64, 100, True, 4),
    ("randtile", 6, 64, 64, 0, True, 4), ("randtile", 8, 96, 96, 50, False, 4),
    ("randtile", 8, 96, 96, 50, True, 3), ("gradn", 9, 96, 56, 80, True, 1),
    ("rand", 10, 256, 256, 95, True, 4), ("randtile", 11, 320, 200, 5, True, 4),
    ("gradn", 12, 100, 60, 50, True, 4),   # W%8 != 0: encode only (decode is UB in the reference, T9)
    ("randtile", 13, 128, 52, 50, True, 4),  # H%8 != 0
])
def test_oracle_equals_real_reference(kind, seed, w, h, q, ycbcr, ch):
    img = himg_amd.synth(kind, seed, w, h)
    if ch != 4:
        img = np.ascontiguousarray(img[:, :, :ch])
    a = ol.oracle_encode(img, q, ycbcr)
    b = ol.ref_encode(img, q, ycbcr)
    assert np.array_equal(a, b)
    if w % 8:
        return
    ra, da = ol.oracle_decode(a)
    rb, db = ol.ref_decode(b)
    assert (ra == 0) == (rb == 0)
    if ra == 0:
        assert np.array_equal(da, db)


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_pixel_stride_larger_than_channels():
    img = himg_amd.synth("randtile", 2, 64, 64)
    a = ol.oracle_encode(img, 50, True, channels=3, stride=4)
    b = ol.ref_encode(img, 50, True, channels=3, stride=4)
    assert np.array_equal(a, b)


def test_oracle_accepts_nonzero_pad_bits():
    """Decoders must ignore the stale pad bits of trap T1."""
    rec = GOLDEN["gradn_s0_64x64_q50"]
    packed = fixture(rec).copy()
    rc, ref_pix = ol.oracle_decode(packed)
    assert rc == 0
    packed[-1] |= 0x80  # top bit of the very last byte is padding in this fixture or data;
    rc2, pix2 = ol.oracle_decode(packed)
    # flipping a pad bit never changes the decode; if it was a data bit the
    # stream may be rejected -- either way the decoder must not crash.
    assert rc2 != 0 or pix2.shape == ref_pix.shape


def test_oracle_under_sanitizers():
    """the restatement under ASan + UBSan (the reference itself trips
    UBSan at quantize.cpp:163 and ASan at decoder.cpp:354; the restatement must be clean).
    The golden 64x64 cases plus ragged shapes go through libhimg_oracle_asan.so in a
    child process (the sanitizer runtime has to be loaded first)."""
    import os
    import subprocess
    import sys
    so = os.path.join(ol.ORACLE_DIR, "libhimg_oracle_asan.so")
    subprocess.run(["make", "-s", "-C", ol.ORACLE_DIR, "libhimg_oracle_asan.so"], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(asan), asan
    code = r'''
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import himg_amd
from golden_util import GOLDEN, cases, fixture, make_input
L = C.CDLL(%r)
L.himg_oracle_free.restype = None
def enc(img, q, ycbcr, ch, st):
    h, w = img.shape[:2]
    out, n = C.POINTER(C.c_uint8)(), C.c_int()
    rc = L.himg_oracle_encode(img.ctypes.data_as(C.c_void_p), w, h, st, ch, q, 1 if ycbcr else 0, C.byref(out), C.byref(n), None)
    assert rc == 0
    a = np.ctypeslib.as_array(out, (n.value,)).copy(); L.himg_oracle_free(out); return a
def dec(p, threads):
    out = C.POINTER(C.c_uint8)(); w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = L.himg_oracle_decode(p.ctypes.data_as(C.c_void_p), p.nbytes, threads, C.byref(out), C.byref(w), C.byref(h), C.byref(c))
    if rc == 0:
        L.himg_oracle_free(out)
    return rc
n = 0
for name in cases(max_pixels=64 * 64):
    rec = GOLDEN[name]
    img = make_input(rec)
    p = enc(img, rec["quality"], bool(rec["ycbcr"]), img.shape[2], img.shape[2])
    assert himg_amd.fnv1a64(p) == rec["stream_fnv"], name
    assert (dec(p, 2) == 0) == bool(rec["decodes"]), name
    n += 1
for (w, h, ch, q) in ((72, 40, 4, 20), (13, 21, 3, 90), (8, 8, 1, 50), (136, 9, 4, 100)):
    img = np.ascontiguousarray(himg_amd.synth("rand", 7, w, h)[:, :, :ch])
    p = enc(img, q, True, ch, ch)
    dec(p, 1)
    bad = p.copy(); bad[len(bad) // 2] ^= 0x55
    dec(bad, 1)          # a damaged stream must not read or write out of bounds either
    n += 1
print("sanitizer run ok:", n, "cases")
''' % (os.path.dirname(ol.ROOT) if False else ol.ROOT, os.path.join(ol.ROOT, "tests"), so)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitizer run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
