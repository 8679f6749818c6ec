"""GPU (-m gpu): the HIP path through the C ABI vs the CPU oracle and the
golden vectors recorded from the real reference.  Bar: bit-exact."""
import numpy as np
import pytest

import himg_amd
import oracle_lib as ol
from golden_util import GOLDEN, cases, fixture, make_input

pytestmark = pytest.mark.gpu


def _eq(a, b, what):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    assert a.size == b.size, "%s: size %d vs %d" % (what, a.size, b.size)
    d = np.nonzero(a != b)[0]
    assert d.size == 0, "%s: %d mismatches, first at %d (gpu=%s oracle=%s)" % (
        what, d.size, d[0], a[d[0]], b[d[0]])


STAGE_CASES = [("gradn", 0, 64, 64, 50), ("rand", 1, 64, 64, 50), ("randtile", 2, 64, 64, 50),
               ("randtile", 3, 200, 120, 50), ("gradn", 4, 512, 512, 50),
               ("randtile", 5, 512, 256, 90), ("randtile", 6, 256, 512, 10),
               ("rand", 7, 136, 72, 100), ("grad", 0, 256, 256, 0)]


@pytest.mark.parametrize("kind,seed,w,h,q", STAGE_CASES)
def test_encode_stages_match_oracle(engine, kind, seed, w, h, q):
    """Every intermediate product of the encoder, not just the final bytes."""
    img = himg_amd.synth(kind, seed, w, h)
    packed_o, tr = ol.oracle_encode(img, q, True, trace=True)
    packed_g = engine.encode(img, q, True)
    n_plane = 4 * tr["rows"] * tr["cols"]
    _eq(engine.debug_read("avg", 0, n_plane), tr["avg"], "box averages")
    _eq(engine.debug_read("lowres", 0, n_plane), tr["lowres"], "low-res plane")
    _eq(engine.debug_read("lres_sym", 0, tr["lres_sym"].size), tr["lres_sym"], "LRES symbols")
    _eq(engine.debug_read("fres_sym", 0, tr["fres_sym"].size), tr["fres_sym"], "FRES symbols")
    for k in ("lres_hist", "fres_hist", "lres_len", "fres_len"):
        _eq(engine.debug_read(k, 0, 261 * 4, np.uint32), tr[k], k)
    for k in ("lres_code", "fres_code"):
        _eq(engine.debug_read(k, 0, 261 * 8, np.uint64), tr[k], k)
    _eq(engine.debug_read("fres_row_bytes", 0, tr["rows"] * 4, np.uint32), tr["fres_row_bytes"],
        "row payload bytes")
    _eq(packed_g, packed_o, "stream")


@pytest.fixture(scope="module")
def unfused_engine():
    """Context that routes block rows through the generic decode path (symbols via
    HBM) -- the path rows wider than the fused kernel's LDS budget always take."""
    import os
    os.environ["HIMG_FORCE_UNFUSED"] = "1"
    try:
        eng = himg_amd.Engine(0)
    finally:
        del os.environ["HIMG_FORCE_UNFUSED"]
    yield eng
    eng.close()


@pytest.mark.parametrize("kind,seed,w,h,q", STAGE_CASES[:6])
def test_decode_stages_match_oracle(engine, unfused_engine, kind, seed, w, h, q):
    img = himg_amd.synth(kind, seed, w, h)
    packed = ol.oracle_encode(img, q, True)
    rc, dt = ol.oracle_decode_trace(packed)
    assert rc == 0
    pix = engine.decode(packed)   # fused path: symbols stay in LDS
    _eq(engine.debug_read("lres_sym", 0, dt["lres_sym"].size, decoder=True), dt["lres_sym"], "LRES symbols")
    _eq(engine.debug_read("lowres", 0, dt["lowres"].size, decoder=True), dt["lowres"], "low-res plane")
    _eq(pix, dt["pixels"], "pixels (fused)")
    pix2 = unfused_engine.decode(packed)
    _eq(unfused_engine.debug_read("fres_sym", 0, dt["fres_sym"].size, decoder=True), dt["fres_sym"], "FRES symbols")
    _eq(pix2, dt["pixels"], "pixels (generic)")


def test_generic_decode_path_large(unfused_engine):
    rec = GOLDEN["randtile_s0_1920x1080_q50"]
    packed = unfused_engine.encode(make_input(rec), 50, True)
    assert himg_amd.fnv1a64(unfused_engine.decode(packed)) == rec["decoded_fnv"]
    rec = GOLDEN["rand_s0_2048x2048_q50"]
    packed = unfused_engine.encode(make_input(rec), 50, True)
    assert himg_amd.fnv1a64(unfused_engine.decode(packed)) == rec["decoded_fnv"]


@pytest.mark.parametrize("name", cases(max_pixels=4096 * 4096))
def test_encode_matches_golden(engine, name):
    """Golden streams from the real reference, up to BASELINE's 4096x4096 configs
    (quality sweep included): size, chunk sizes and FNV-1a-64 of the bytes."""
    rec = GOLDEN[name]
    img = make_input(rec)
    packed = engine.encode(img, rec["quality"], bool(rec["ycbcr"]))
    assert packed.size == rec["packed_size"]
    assert himg_amd.fnv1a64(packed) == rec["stream_fnv"]
    if "fixture" in rec:
        _eq(packed, fixture(rec), "fixture bytes")


@pytest.mark.parametrize("name", cases(max_pixels=4096 * 4096))
def test_decode_matches_golden(engine, name):
    rec = GOLDEN[name]
    img = make_input(rec)
    packed = fixture(rec) if "fixture" in rec else engine.encode(img, rec["quality"], bool(rec["ycbcr"]))
    if not rec["decodes"]:
        with pytest.raises(himg_amd.HimgError) as e:   # trap T2: same accept/reject as the reference
            engine.decode(packed)
        assert e.value.code == himg_amd.HIMG_ERR_FORMAT
        return
    dec = engine.decode(packed)
    assert dec.shape == (rec["height"], rec["width"], rec["channels"])
    assert himg_amd.fnv1a64(dec) == rec["decoded_fnv"]
    assert round(himg_amd.psnr(img, dec), 4) == pytest.approx(rec["psnr"], abs=1e-4)


@pytest.mark.parametrize("w,h,ch,stride,ycbcr,q", [
    (64, 64, 3, 3, True, 50), (64, 64, 3, 4, True, 50), (72, 40, 1, 1, True, 50),
    (72, 40, 2, 2, True, 70), (64, 64, 4, 4, False, 50), (128, 52, 4, 4, True, 50),
    (100, 60, 4, 4, True, 50), (8, 8, 4, 4, True, 50), (8, 200, 4, 4, True, 50),
    (520, 24, 4, 4, True, 30), (4104, 16, 4, 4, True, 50),
    # W % 8 != 0 (the reference decoder is undefined there, trap T9; ours clips, see below)
    (97, 33, 4, 4, True, 50), (1, 1, 4, 4, True, 50), (13, 200, 3, 3, True, 70), (250, 9, 1, 1, True, 50),
    (4099, 24, 4, 4, True, 50), (67, 67, 4, 4, False, 90)])
def test_general_shapes_match_oracle(engine, w, h, ch, stride, ycbcr, q):
    """Channel counts, pixel_stride > channels, -rgb mode, ragged heights and
    widths, single-block-row images (no size headers)."""
    img = himg_amd.synth("randtile", w * 7 + h, w, h)
    if stride != 4:
        img = np.ascontiguousarray(img[:, :, :stride])
    a = engine.encode(img, q, ycbcr, channels=ch, pixel_stride=stride)
    b = ol.oracle_encode(img, q, ycbcr, channels=ch, stride=stride)
    _eq(a, b, "stream")
    rc, pix = ol.oracle_decode(b)
    if rc != 0:
        with pytest.raises(himg_amd.HimgError):
            engine.decode(b)
    else:
        # W % 8 != 0: the reference decoder reads and writes out of bounds
        # (decoder.cpp:63-72, trap T9).  The DEFINED behaviour here: a partial tile is
        # decoded like a full one and only its columns inside the image are stored
        # (what the encoder's edge replication, encoder.cpp:26-52, calls for); the
        # oracle restates exactly that.
        _eq(engine.decode(b), pix, "pixels")


def test_long_zero_runs_and_run_splitting(engine):
    """Flat and nearly flat frames: runs longer than 16662 are split greedily from
    the run start and never cross a block row (trap T6); LRES spans are all zero."""
    for w, h, fill in [(2048, 64, 7), (1024, 1024, 200)]:
        img = np.full((h, w, 4), fill, np.uint8)
        img[h // 2, w // 3] = 255 - fill
        a = engine.encode(img, 50, True)
        b = ol.oracle_encode(img, 50, True)
        _eq(a, b, "stream %dx%d" % (w, h))


def test_device_batch_api(engine):
    """Batched, HBM-resident entry points: independent frames in one launch each
    equal their single-frame golden streams; batched decode returns the pixels."""
    import torch
    names = ["randtile_s0_1920x1080_q50", "randtile_s1_1920x1080_q50", "randtile_s255_1920x1080_q50"]
    recs = [GOLDEN[n] for n in names]
    w, h = 1920, 1080
    frames = np.stack([make_input(r) for r in recs] + [himg_amd.synth("rand", 9, w, h)])
    B = frames.shape[0]
    dev = torch.device("cuda:0")
    d_frames = torch.from_numpy(frames).to(dev)
    cap = himg_amd.max_packed_size(w, h, 4)
    d_out = torch.empty((B, cap), dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(B, dtype=torch.int32, device=dev)
    d_status = torch.full((B,), -1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    engine.encode_device(d_frames, B, w, h, 4, 4, 50, True, d_out, cap, d_sizes, d_status, stream)
    torch.cuda.synchronize()
    sizes = d_sizes.cpu().numpy().astype(np.int64)
    assert not d_status.cpu().numpy().any()
    out = d_out.cpu().numpy()
    for i, r in enumerate(recs):
        assert sizes[i] == r["packed_size"]
        assert himg_amd.fnv1a64(out[i, :sizes[i]]) == r["stream_fnv"]
    _eq(out[3, :sizes[3]], ol.oracle_encode(frames[3], 50, True), "frame 3")

    d_pix = torch.empty((B, h, w, 4), dtype=torch.uint8, device=dev)
    d_st2 = torch.full((B,), -1, dtype=torch.int32, device=dev)
    engine.decode_device(d_out, cap, sizes.astype(np.uint32), B, w, h, 4, d_pix, d_st2, stream)
    torch.cuda.synchronize()
    assert not d_st2.cpu().numpy().any()
    pix = d_pix.cpu().numpy()
    for i, r in enumerate(recs):
        assert himg_amd.fnv1a64(pix[i]) == r["decoded_fnv"]


def test_decoder_rejects_like_reference(engine):
    rec = GOLDEN["gradn_s0_64x64_q50"]
    good = fixture(rec)
    # Not RIFF / wrong total size / truncated / damaged row header.
    bad = [good.copy() for _ in range(4)]
    bad[0][0] = ord("X")
    bad[1] = np.concatenate([good, np.zeros(3, np.uint8)])
    bad[2] = good[:-40].copy()
    bad[2][4:8] = np.frombuffer(np.uint32(bad[2].size - 8).tobytes(), np.uint8)
    for b in bad[:3]:
        rc, _ = ol.oracle_decode(b)
        assert rc != 0
        with pytest.raises(himg_amd.HimgError):
            engine.decode(b)
    # Stale pad bits (trap T1) are ignored by the decoder.
    _eq(engine.decode(good), ol.oracle_decode(good)[1], "pixels")


def test_engine_is_reusable_and_fresh_per_call(engine):
    """Every encode has fresh-Encoder semantics (trap T4) and contexts are reusable."""
    a = himg_amd.synth("randtile", 1, 128, 128)
    b = himg_amd.synth("gradn", 2, 256, 64)
    e1 = engine.encode(a, 50)
    engine.encode(b, 30)
    e2 = engine.encode(a, 50)
    _eq(e1, e2, "repeat encode")
    _eq(e1, ol.oracle_encode(a, 50), "oracle")


def test_host_buffer_variants(engine):
    """himg_hip_encode_to / himg_hip_fetch_last / himg_hip_decode_to: caller-owned
    buffers, HIMG_ERR_CAPACITY with the needed size, results identical to the
    allocating entry points."""
    import ctypes as C
    L = himg_amd.lib()
    img = himg_amd.synth("randtile", 3, 320, 200)
    want = ol.oracle_encode(img, 50, True)
    ctx = engine._ctx
    n = C.c_size_t()
    small = np.empty(16, np.uint8)
    rc = L.himg_hip_encode_to(ctx, img.ctypes.data, 320, 200, 4, 4, 50, 1, small.ctypes.data, small.nbytes, C.byref(n))
    assert rc == himg_amd.HIMG_ERR_CAPACITY and n.value == len(want)
    buf = np.empty(n.value + 7, np.uint8)
    assert L.himg_hip_fetch_last(ctx, buf.ctypes.data, buf.nbytes, C.byref(n)) == 0
    _eq(buf[: n.value], want, "fetched stream")
    big = np.empty(himg_amd.max_packed_size(320, 200, 4), np.uint8)
    assert L.himg_hip_encode_to(ctx, img.ctypes.data, 320, 200, 4, 4, 50, 1, big.ctypes.data, big.nbytes, C.byref(n)) == 0
    _eq(big[: n.value], want, "stream in the caller's buffer")
    _eq(engine.encode(img, 50, True), want, "Engine.encode")

    rc0, pix = ol.oracle_decode(want)
    assert rc0 == 0
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = L.himg_hip_decode_to(ctx, want.ctypes.data, want.nbytes, None, 0, C.byref(w), C.byref(h), C.byref(c))
    assert rc == himg_amd.HIMG_ERR_CAPACITY and (w.value, h.value, c.value) == (320, 200, 4)
    out = np.empty(320 * 200 * 4, np.uint8)
    assert L.himg_hip_fetch_last(ctx, out.ctypes.data, out.nbytes, C.byref(n)) == 0 and n.value == out.nbytes
    _eq(out, pix.ravel(), "fetched pixels")
    reuse = np.zeros((200, 320, 4), np.uint8)
    got = engine.decode(want, out=reuse)
    assert got.ctypes.data == reuse.ctypes.data   # decoded in place, no new allocation
    _eq(got.ravel(), pix.ravel(), "pixels in the caller's buffer")


def test_emit_staging_overflow_windows(engine):
    """k_emit stages an iteration's bits (4096 symbols) in a 4 KiB circular buffer
    and falls back to several windows when they do not fit (> 32 640 bits).  A
    frame of two-valued tiles with one band of random tiles gives that band's
    low-res deltas rare, 11..19-bit codes over 8192 consecutive LRES symbols."""
    rng = np.random.default_rng(5)
    w, h, band = 1024, 4096, 512
    a = rng.integers(0, 2, (h // 8, w // 8, 4), dtype=np.uint8) * 16 + 120
    a[: band // 8] = rng.integers(0, 256, (band // 8, w // 8, 4), dtype=np.uint8)
    img = np.ascontiguousarray(np.repeat(np.repeat(a, 8, axis=0), 8, axis=1))
    want, tr = ol.oracle_encode(img, 50, True, trace=True)
    sym = np.array(tr["lres_sym"]).ravel()
    lens = np.array(tr["lres_len"]).astype(np.int64)
    cost = lens[sym] * (sym != 0)   # literal bits only: a lower bound of the iteration's bits
    worst = max(int(cost[i:i + 4096].sum()) for s0 in range(0, sym.size, 16384)
                for i in range(s0, min(s0 + 16384, sym.size), 4096))
    assert worst > 1020 * 32, "the input no longer overflows the staging window"
    assert lens.max() > 11   # codes longer than the decoder's first-level table, too
    _eq(engine.encode(img, 50, True), want, "stream")
    rc, pix = ol.oracle_decode(want)
    assert rc == 0
    _eq(engine.decode(want).ravel(), pix.ravel(), "pixels")


def _chunks(stream):
    import struct
    b, out, i = bytes(stream), {}, 12
    while i + 8 <= len(b):
        sz = struct.unpack("<I", b[i + 4:i + 8])[0]
        out[b[i:i + 4].decode()] = (i + 8, sz)
        i += 8 + sz
    return out


@pytest.mark.parametrize("kind,w,h,q", [("randtile", 256, 128, 90), ("gradn", 512, 512, 50), ("rand", 128, 64, 50)])
def test_bit_flips_in_the_payload_decode_like_the_reference(engine, unfused_engine, kind, w, h, q):
    """Streams no encoder produced: one flipped bit in the LRES / FRES payload (or,
    sometimes, in the serialised tree).  The container stays valid, so the
    reference's behaviour is defined: it either rejects the stream or decodes
    different pixels -- and the GPU decoder (fused and generic path) must do exactly
    the same, bit for bit."""
    img = himg_amd.synth(kind, 3, w, h)
    good = ol.oracle_encode(img, q, True)
    assert ol.oracle_decode(good)[0] == 0
    ch = _chunks(good)
    rng = np.random.default_rng(1)
    accepted = rejected = 0
    for t in range(120):
        bad = good.copy()
        off, sz = ch["FRES" if t % 3 else "LRES"]
        lo = off + (0 if t % 10 == 0 else min(400, sz // 2))
        bad[int(rng.integers(lo, off + sz))] ^= 1 << int(rng.integers(0, 8))
        rc, pix = ol.oracle_decode(bad)
        for eng in (engine, unfused_engine) if t % 4 == 0 else (engine,):
            if rc == 0:
                _eq(eng.decode(bad).ravel(), pix.ravel(), "pixels of mutation %d" % t)
            else:
                with pytest.raises(himg_amd.HimgError):
                    eng.decode(bad)
        accepted += rc == 0
        rejected += rc != 0
    assert accepted > 10 and rejected > 10   # the mutations exercise both outcomes


def test_damaged_trees_decode_like_the_reference(engine):
    """The serialised Huffman trees of the LRES and FRES chunks (k_dec_parse recovers
    them in parallel: node starts per 32-bit word, then the tree from prefix sums and
    pointer jumping) with one to three flipped bits: a leaf turned into a branch shifts
    every later node, trees end early or run into the payload, symbols change.  The
    reference either rejects such a stream or decodes other pixels (huffman_dec.cpp:152-229);
    the GPU decoder must agree, bit for bit."""
    img = himg_amd.synth("rand", 11, 512, 128)
    good = ol.oracle_encode(img, 50, True)
    assert ol.oracle_decode(good)[0] == 0
    ch = _chunks(good)
    rng = np.random.default_rng(5)
    accepted = rejected = 0
    for t in range(400):
        bad = good.copy()
        off, sz = ch["FRES" if t % 2 else "LRES"]
        for _ in range(1 + t % 3):
            bad[off + int(rng.integers(0, min(sz, 340)))] ^= 1 << int(rng.integers(0, 8))
        rc, pix = ol.oracle_decode(bad)
        if rc == 0:
            _eq(engine.decode(bad).ravel(), pix.ravel(), "pixels of mutation %d" % t)
        else:
            with pytest.raises(himg_amd.HimgError):
                engine.decode(bad)
        accepted += rc == 0
        rejected += rc != 0
    assert accepted > 10 and rejected > 10


def test_row_ranges_decode_like_one_piece():
    """HIMG_WALK_SEGS (by default only frames of 1024 block rows or more: BASELINE config 4):
    the row-header walk in up to four launches, the counts and the row kernels of a range
    beside the walk of the next.  Good streams, streams whose damage sits in a later range
    (rows of the earlier ranges are decoded before the walk finds it) and truncated ones
    must come out exactly as from the one-piece decode -- fused and generic row kernels."""
    import os
    engs = []
    try:
        for segs, unfused in ((4, False), (3, True)):
            os.environ["HIMG_WALK_SEGS"] = str(segs)
            if unfused:
                os.environ["HIMG_FORCE_UNFUSED"] = "1"
            try:
                engs.append(himg_amd.Engine(0))
            finally:
                del os.environ["HIMG_WALK_SEGS"]
                os.environ.pop("HIMG_FORCE_UNFUSED", None)
        rng = np.random.default_rng(9)
        for kind, w, h, q in (("rand", 512, 1024, 50), ("randtile", 1024, 640, 80), ("gradn", 200, 900, 60)):
            img = himg_amd.synth(kind, 2, w, h)
            good = ol.oracle_encode(img, q, True)
            rc, pix = ol.oracle_decode(good)
            assert rc == 0
            ch = _chunks(good)
            off, sz = ch["FRES"]
            offs = himg_amd.index_host(good)[3]
            for eng in engs:
                _eq(eng.decode(good).ravel(), pix.ravel(), "%s %dx%d" % (kind, w, h))
            accepted = rejected = 0
            for t in range(40):
                bad = good.copy()
                lo = off + sz // 4 * (t % 4)          # the damage in each quarter of the chunk in turn
                bad[int(rng.integers(lo, lo + sz // 4))] ^= 1 << int(rng.integers(0, 8))
                if t % 8 == 7:
                    bad[off + sz - 3] ^= 0x55          # and near its end
                if t % 5 == 4:                         # a damaged size header of a row in that quarter
                    r = min(len(offs) - 1, len(offs) // 4 * (t % 4) + int(rng.integers(0, len(offs) // 4)))
                    bad[int(offs[r]) - 2] ^= 1 << int(rng.integers(0, 8))
                rc, pix_bad = ol.oracle_decode(bad)
                for eng in engs:
                    if rc == 0:
                        _eq(eng.decode(bad).ravel(), pix_bad.ravel(), "mutation %d" % t)
                    else:
                        with pytest.raises(himg_amd.HimgError):
                            eng.decode(bad)
                accepted += rc == 0
                rejected += rc != 0
            assert accepted > 0 and rejected > 3, (accepted, rejected)
    finally:
        for e in engs:
            e.close()


def test_fixed_mode_decodes_what_the_reference_cannot(engine):
    """HIMG_OPT_FIX_T2 (opt-in): streams of the reference's own encoder that its
    decoder rejects -- compressible frames (trap T2), frames of one block row,
    one-symbol token alphabets -- decode, bit-identical to the oracle with the same
    rules switched on; everything else decodes exactly as before; the default stays
    the reference's behaviour."""
    cases = [("flat", np.full((64, 256, 4), 77, np.uint8)), ("flat-large", np.full((1024, 1024, 4), 200, np.uint8)),
             ("grad", himg_amd.synth("grad", 0, 512, 512)), ("one block row", himg_amd.synth("randtile", 1, 256, 8)),
             ("ordinary", himg_amd.synth("randtile", 1, 256, 64))]
    fixed = himg_amd.Engine(0)
    fixed.set_option("fix_t2", 1)
    try:
        for name, img in cases:
            packed = ol.oracle_encode(img, 50, True)
            rc_ref, pix_ref = ol.oracle_decode(packed)
            rc_fix, pix_fix = ol.oracle_decode(packed, fix_t2=True)
            assert rc_fix == 0, name
            got = fixed.decode(packed)
            _eq(got.ravel(), pix_fix.ravel(), "fixed-mode pixels of " + name)
            assert himg_amd.psnr(img, got) > 30.0
            if rc_ref == 0:
                _eq(pix_ref.ravel(), pix_fix.ravel(), "fixed mode changes nothing for " + name)
                _eq(engine.decode(packed).ravel(), pix_ref.ravel(), "default-mode pixels of " + name)
            else:
                with pytest.raises(himg_amd.HimgError):
                    engine.decode(packed)
    finally:
        fixed.close()


def test_batched_host_api_frames_in_flight(engine):
    """himg_hip_encode_batch / himg_hip_decode_batch: the pipelined host path gives
    the same bytes and pixels as frame-by-frame calls; a bad stream in the middle
    fails alone (mixed geometries in one decode call)."""
    frames = [himg_amd.synth("randtile", s, 640, 360) for s in range(5)]
    want = [ol.oracle_encode(f, 50, True) for f in frames]
    got = engine.encode_batch(frames, 50, True)
    assert len(got) == 5
    for g_, w_ in zip(got, want):
        _eq(g_, w_, "stream")
    # decode: different geometries, one T2 stream (rejected), outputs reused
    small = ol.oracle_encode(himg_amd.synth("gradn", 1, 128, 72), 90, True)
    t2 = ol.oracle_encode(himg_amd.synth("grad", 0, 512, 512), 50, True)
    assert ol.oracle_decode(t2)[0] != 0
    streams = [want[0], small, want[1]]
    pix = engine.decode_batch(streams)
    for p_, s_ in zip(pix, streams):
        _eq(p_.ravel(), ol.oracle_decode(s_)[1].ravel(), "pixels")
    import ctypes as C
    L = himg_amd.lib()
    streams = [want[0], t2, want[2], small]
    n = len(streams)
    outs = [np.zeros(640 * 360 * 4, np.uint8) for _ in range(n)]
    src = (C.c_void_p * n)(*[s_.ctypes.data for s_ in streams])
    szs = (C.c_size_t * n)(*[s_.nbytes for s_ in streams])
    dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
    ws, hs, cs = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
    rc = L.himg_hip_decode_batch(engine._ctx, src, szs, n, dst, caps, ws, hs, cs)
    assert rc == himg_amd.HIMG_ERR_FORMAT            # the first (only) failure
    assert (ws[1], hs[1], cs[1]) == (0, 0, 0)       # the T2 stream
    for i in (0, 2, 3):
        ref = ol.oracle_decode(streams[i])[1]
        assert (hs[i], ws[i], cs[i]) == ref.shape
        _eq(outs[i][: ref.size], ref.ravel(), "pixels of frame %d" % i)


def test_streams_longer_than_one_chunk():
    """A stream whose payload exceeds 1024 sub-sequences is decoded chunk after chunk
    (each chunk's first token is exact once the previous chunk has finished).  With
    sub-sequences of up to 4096 bits that takes a > 4 Mbit block row, so the test
    lowers the cap (HIMG_MAX_SUB_BITS, a test knob) to 128 / 256 bits: ordinary
    rows then need several chunks -- in the fused row kernel (k_row_count steps
    aside), in the generic path, and in the serial LRES fallback."""
    import os
    img = himg_amd.synth("randtile", 6, 4096, 64)
    img2 = himg_amd.synth("rand", 2, 1024, 128)
    cases = [(img, 90), (img, 50), (img2, 50)]
    for cap, unfused in ((128, False), (256, False), (256, True)):
        os.environ["HIMG_MAX_SUB_BITS"] = str(cap)
        if unfused:
            os.environ["HIMG_FORCE_UNFUSED"] = "1"
        try:
            eng = himg_amd.Engine(0)
        finally:
            del os.environ["HIMG_MAX_SUB_BITS"]
            os.environ.pop("HIMG_FORCE_UNFUSED", None)
        try:
            for im, q in cases:
                packed = ol.oracle_encode(im, q, True)
                rc, pix = ol.oracle_decode(packed)
                assert rc == 0
                _eq(eng.decode(packed).ravel(), pix.ravel(), "pixels (cap %d, unfused %s, q %d)" % (cap, unfused, q))
                st = eng.debug_read("dec_stats", 0, ((im.shape[0] + 7) // 8 + 1) * 32, np.uint32, decoder=True)
                if q == 90 or cap == 128:   # these certainly exceed 1024 sub-sequences per row
                    assert st.reshape(-1, 8)[1:, 0].max() > 1, "no block row needed more than one chunk"
        finally:
            eng.close()


@pytest.mark.parametrize("lead,side", [(0, "1"), (32, "1"), (200, "0"), (4096, "1")])
def test_lead_in_and_side_stream_do_not_change_results(lead, side):
    """The lead-in of the speculative sub-sequence decode (HIMG_LEAD_BITS) and the
    side stream (HIMG_SIDE_STREAM) are pure scheduling: every setting decodes the
    oracle's streams to the oracle's pixels, on the fused and the generic path,
    mutated streams included."""
    import os
    os.environ["HIMG_LEAD_BITS"] = str(lead)
    os.environ["HIMG_SIDE_STREAM"] = side
    try:
        eng = himg_amd.Engine(0)
    finally:
        del os.environ["HIMG_LEAD_BITS"]
        del os.environ["HIMG_SIDE_STREAM"]
    try:
        rng = np.random.default_rng(lead)
        for kind, w, h, q in [("randtile", 4096, 48, 50), ("rand", 520, 72, 90), ("randtile", 4400, 40, 70),
                              ("gradn", 1920, 136, 50)]:
            good = ol.oracle_encode(himg_amd.synth(kind, 3, w, h), q, True)
            rc, pix = ol.oracle_decode(good)
            if rc != 0:
                continue   # trap T2: the reference rejects its own stream
            _eq(eng.decode(good).ravel(), pix.ravel(), "pixels %dx%d lead %d" % (w, h, lead))
            fres = _chunks(good)["FRES"]
            for _ in range(6):
                bad = good.copy()
                bad[int(rng.integers(fres[0], fres[0] + fres[1]))] ^= 1 << int(rng.integers(0, 8))
                rc, pix = ol.oracle_decode(bad)
                if rc == 0:
                    _eq(eng.decode(bad).ravel(), pix.ravel(), "pixels of a mutated stream, lead %d" % lead)
                else:
                    with pytest.raises(himg_amd.HimgError):
                        eng.decode(bad)
    finally:
        eng.close()


def test_lres_serial_fallback():
    """If a mis-speculation runs through a whole LRES chunk the parallel chunk chain
    does not verify and the frame's LRES stream is decoded by one workgroup instead
    (k_dec_huff).  HIMG_FORCE_LRES_SERIAL=1 (a test knob) takes that path on purpose;
    with HIMG_MAX_SUB_BITS=256 the serial stream also needs several chunks."""
    import os
    os.environ["HIMG_FORCE_LRES_SERIAL"] = "1"
    os.environ["HIMG_MAX_SUB_BITS"] = "256"
    try:
        eng = himg_amd.Engine(0)
    finally:
        del os.environ["HIMG_FORCE_LRES_SERIAL"]
        del os.environ["HIMG_MAX_SUB_BITS"]
    try:
        for kind, w, h, q in [("randtile", 2048, 1024, 50), ("gradn", 640, 360, 90), ("randtile", 64, 64, 100)]:
            packed = ol.oracle_encode(himg_amd.synth(kind, 8, w, h), q, True)
            rc, pix = ol.oracle_decode(packed)
            assert rc == 0
            _eq(eng.decode(packed).ravel(), pix.ravel(), "pixels %dx%d" % (w, h))
            st = eng.debug_read("dec_stats", 0, ((h + 7) // 8 + 1) * 32, np.uint32, decoder=True).reshape(-1, 8)
            # Row 0 of the statistics is the LRES stream; only the serial decode fills in
            # the payload and output sizes (slots 6, 7).
            assert st[0, 7] > 0 and st[0, 6] > 0, "the serial LRES path did not run"
            if w == 2048:
                assert st[0, 0] > 1, "the serial LRES decode did not need more than one chunk"
        bad = ol.oracle_encode(himg_amd.synth("randtile", 8, 640, 360), 70, True)
        lres = _chunks(bad)["LRES"]
        bad[lres[0] + lres[1] // 2] ^= 0x10
        rc, pix = ol.oracle_decode(bad)
        if rc == 0:
            _eq(eng.decode(bad).ravel(), pix.ravel(), "pixels of the mutated stream")
        else:
            with pytest.raises(himg_amd.HimgError):
                eng.decode(bad)
    finally:
        eng.close()
