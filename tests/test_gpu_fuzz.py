"""GPU (-m gpu): seeded fuzzing of the decoder against the oracle -- streams no encoder
produced (the reference's parser has TODOs exactly there: huffman_dec.cpp:114,135,244).
Every mutation keeps the RIFF container valid, so the reference's behaviour is defined:
it rejects the stream or decodes other pixels, and the GPU decoder must do the same, bit
for bit -- with BOTH forms of the FRES row index kernel (a workgroup per row /
a wavefront per row, HIMG_OPT_COUNT_WAVE), which divide a row's payload among the lanes
differently.  Bounded: a few thousand streams, well under a minute.  The long form
(tools/fuzz_decode.py, nine geometries up to 16384 pixels wide) is logged in
profiles/r04_fuzz_decode.txt."""
import struct
import time

import numpy as np
import pytest

import himg_amd
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _chunks(stream):
    b, out, i = bytes(stream), {}, 12
    while i + 8 <= len(b):
        sz = struct.unpack("<I", b[i + 4:i + 8])[0]
        out[b[i:i + 4].decode()] = (i + 8, sz)
        i += 8 + sz
    return out


def _mutate(good, ch, rng, t):
    """One hostile stream: flipped bits in a payload (t % 4 in 0, 2), in a serialised tree
    (1), in a table or a row size header (3); sometimes a whole byte replaced."""
    bad = good.copy()
    kind = t % 4
    if kind == 1:
        off, sz = ch["FRES" if t % 8 == 1 else "LRES"]
        for _ in range(1 + t % 3):
            bad[off + int(rng.integers(0, min(sz, 340)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 3:
        which = ("FMAP", "QCFG", "LMAP", "FRES")[(t // 4) % 4]
        off, sz = ch[which]
        if which == "FRES":      # the neighbourhood of the first row headers
            lo = min(sz - 1, 330)
            bad[off + int(rng.integers(lo, min(sz, lo + 4096)))] ^= 1 << int(rng.integers(0, 8))
        else:
            bad[off + int(rng.integers(0, sz))] ^= 1 << int(rng.integers(0, 8))
    else:
        off, sz = ch["FRES" if kind == 0 else "LRES"]
        lo = off + min(400, sz // 2)
        n = 1 + (t % 5 == 0)
        for _ in range(n):
            i = int(rng.integers(lo, off + sz))
            if t % 11 == 0:
                bad[i] = int(rng.integers(0, 256))
            else:
                bad[i] ^= 1 << int(rng.integers(0, 8))
    return bad


BASES = [("randtile", 4096, 64, 50), ("rand", 512, 128, 50), ("gradn", 1024, 256, 70), ("randtile", 200, 116, 90),
         ("randtile", 4400, 40, 50)]
PER_BASE = 220          # x 5 geometries x 2 kernel forms = 2200 streams


@pytest.mark.parametrize("count_wave", [0, 1])
def test_fuzz_decode_matches_oracle(count_wave):
    eng = himg_amd.Engine(0)
    eng.set_option("count_wave", count_wave)
    rng = np.random.default_rng(2024 + count_wave)
    t0 = time.time()
    accepted = rejected = 0
    for kind, w, h, q in BASES:
        img = himg_amd.synth(kind, 5, w, h)
        good = ol.oracle_encode(img, q, True)
        rc, pix = ol.oracle_decode(good)
        assert rc == 0
        assert np.array_equal(eng.decode(good).ravel(), pix.ravel())
        ch = _chunks(good)
        for t in range(PER_BASE):
            bad = _mutate(good, ch, rng, t)
            rc, pix = ol.oracle_decode(bad)
            try:
                got = eng.decode(bad)
            except himg_amd.HimgError:
                got = None
            assert (rc == 0) == (got is not None), "%s %dx%d q%d mutation %d: oracle rc %d, gpu %s" % (
                kind, w, h, q, t, rc, "accepted" if got is not None else "rejected")
            if rc == 0:
                assert np.array_equal(got.ravel(), pix.ravel()), "%s %dx%d q%d mutation %d: pixels differ" % (kind, w, h, q, t)
            accepted += rc == 0
            rejected += rc != 0
    eng.close()
    assert accepted + rejected == PER_BASE * len(BASES)
    assert accepted > 100 and rejected > 100      # the mutations exercise both outcomes
    assert time.time() - t0 < 60, "the bounded fuzz run must stay under a minute"


@pytest.mark.parametrize("emit_rows,count_wave", [(0, 1), (1, 0)])
def test_kernel_forms_give_identical_results(emit_rows, count_wave):
    """The wavefront-per-row forms of k_emit / k_row_count are chosen by launch size; forced
    either way on frames of either size the stream and the pictures are the oracle's."""
    eng = himg_amd.Engine(0)
    eng.set_option("emit_rows", emit_rows)
    eng.set_option("count_wave", count_wave)
    for kind, w, h, q in [("randtile", 4096, 128, 50), ("gradn", 520, 264, 90), ("rand", 256, 64, 10)]:
        img = himg_amd.synth(kind, 9, w, h)
        want = ol.oracle_encode(img, q, True)
        assert np.array_equal(eng.encode(img, q, True), want)
        rc, pix = ol.oracle_decode(want)
        if rc == 0:
            assert np.array_equal(eng.decode(want).ravel(), pix.ravel())
        else:
            with pytest.raises(himg_amd.HimgError):
                eng.decode(want)
    eng.close()


WIDE_BASES = [("randtile", 4400, 40, 50), ("rand", 4608, 24, 30), ("randtile", 8192, 64, 50)]


def test_fuzz_wide_rows_quarter_count_kernel():
    """Rows wider than the LDS with the DEFAULT choice of the count kernel: k_row_count_q (4096
    sub-sequences per row, staged reader, sixteen wavefronts that agree on the chain afterwards)
    -- the forced forms above are the other two kernels.  Hostile streams must get the oracle's
    verdict and pixels here as well."""
    eng = himg_amd.Engine(0)
    assert eng.get_option("count_wave") == -1
    rng = np.random.default_rng(77)
    accepted = rejected = 0
    for kind, w, h, q in WIDE_BASES:
        img = himg_amd.synth(kind, 5, w, h)
        good = ol.oracle_encode(img, q, True)
        rc, pix = ol.oracle_decode(good)
        assert rc == 0
        assert np.array_equal(eng.decode(good).ravel(), pix.ravel()), (kind, w, h, q)
        ch = _chunks(good)
        for t in range(120):
            bad = _mutate(good, ch, rng, t)
            rc, pix = ol.oracle_decode(bad)
            try:
                got = eng.decode(bad)
            except himg_amd.HimgError:
                got = None
            assert (rc == 0) == (got is not None), "%s %dx%d q%d mutation %d: oracle rc %d, gpu %s" % (
                kind, w, h, q, t, rc, "accepted" if got is not None else "rejected")
            if rc == 0:
                assert np.array_equal(got.ravel(), pix.ravel()), "%s %dx%d q%d mutation %d: pixels differ" % (kind, w, h, q, t)
            accepted += rc == 0
            rejected += rc != 0
    eng.close()
    assert accepted > 30 and rejected > 30


def test_wide_rows_beyond_the_staged_reader_fall_back():
    """A frame whose MEAN bit rate lets the host pick k_row_count_q while some of its rows do
    not fit that kernel's staging buffer (a smooth gradient above, random tiles at q = 90 in the
    last two block rows): the kernel leaves those rows' records unusable and the general
    decoder takes them -- same pixels as the oracle.  (Full-swing noise at that quality is no
    test input: the reference encoder itself overruns its output buffer on it.)"""
    w, h = 16384, 128
    img = himg_amd.synth("grad", 2, w, h).copy()
    img[112:] = himg_amd.synth("randtile", 3, w, h)[112:]
    good = ol.oracle_encode(img, 90, True)
    rc, pix = ol.oracle_decode(good)
    assert rc == 0
    # the premise: the mean says "staged reader" (<= 272 bits per quarter sub-sequence), the
    # busy rows do not fit it (> 320)
    rows = h // 8
    assert 8.0 * good.size / (rows * 4096.0) <= 272.0, good.size
    _, _, _, off, ln, first = himg_amd.index_host(good)
    assert (np.asarray(ln[-2:]) * 8 / 4096.0 > 320).all(), ln[-2:]
    eng = himg_amd.Engine(0)
    assert np.array_equal(eng.decode(good).ravel(), pix.ravel())
    eng.close()
