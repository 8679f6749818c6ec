"""GPU (-m gpu): FRES rows through the token stream (k_tok -> k_emit_tok, what batches take)
against the CPU oracle: final bytes, the slots expanded into symbols again, the histogram --
on shapes and contents that exercise every kind of slot (literals behind 0..255 zeros, runs on
their own incl. the reference's greedy split at 16 662, trailing zeros of a row, dense
iterations staged in two parts) and both paths of the bit packer.  Bar: bit-exact."""
import numpy as np
import pytest

import himg_amd
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _images(w, h):
    yield "randtile", himg_amd.synth("randtile", 1, w, h)
    yield "rand", himg_amd.synth("rand", 3, w, h)              # dense: iterations beyond the staging buffer
    z = np.zeros((h, w, 4), np.uint8)
    yield "zero", z                                             # a row is ONE run: the split at 16 662
    s = z.copy()
    s[h // 2:, :, :] = himg_amd.synth("rand", 4, w, h)[h // 2:, :, :]
    s[: h // 2, w - 9, 1] = 200                                 # a lone column of detail: runs of hundreds of zeros
    yield "sparse", s


SHAPES = [(512, 64), (1024, 72), (4096, 64), (1920, 136), (200, 72), (2048, 40)]


@pytest.mark.parametrize("w,h", SHAPES)
def test_token_stream_encode_matches_oracle(w, h):
    eng = himg_amd.Engine(0)
    for name, img in _images(w, h):
        for q, ycc in ((50, True), (90, False), (100, True), (10, True)):
            want, tr = ol.oracle_encode(img, q, ycc, trace=True)
            for mode in (1, 2):   # 2: the bit packer's spelled-out path on every step
                eng.set_option("row_tokens", mode)
                got = eng.encode(img, q, ycc)
                assert got.size == want.size and np.array_equal(got, want), (name, q, ycc, mode)
            sym = eng.debug_read("fres_tok_sym", 0, tr["fres_sym"].size)
            assert np.array_equal(sym, tr["fres_sym"]), (name, q, ycc, "slots expanded")
            assert np.array_equal(eng.debug_read("fres_hist", 0, 261 * 4, np.uint32), tr["fres_hist"]), (name, q, ycc)
    eng.close()


def test_token_stream_other_channel_counts():
    """1-3 channels and a pixel stride beyond the channel count go through k_tile_fwd, then the same slots."""
    eng = himg_amd.Engine(0)
    eng.set_option("row_tokens", 1)
    base = himg_amd.synth("randtile", 7, 264, 80)
    for ch, stride in ((1, 1), (2, 2), (3, 3), (3, 4), (1, 4)):
        img = np.ascontiguousarray(base[:, :, :stride])
        want = ol.oracle_encode(img, 50, True, channels=ch, stride=stride)
        got = eng.encode(img, 50, True, channels=ch, pixel_stride=stride)
        assert np.array_equal(got, want), (ch, stride)
    eng.close()


def test_token_stream_is_what_batches_take():
    """The default (-1) chooses the slots by launch size: a batch of >= 8192 block rows takes them, a
    single frame does not; either way the streams are the oracle's."""
    import torch
    eng = himg_amd.Engine(0)
    assert eng.get_option("row_tokens") == -1
    w, h, B = 1024, 512, 128                       # 64 rows x 128 frames = 8192 block rows
    frames = np.stack([himg_amd.synth("randtile", s, w, h) for s in range(B)])
    d_frames = torch.from_numpy(frames).cuda()
    cap = himg_amd.max_packed_size(w, h, 4)
    d_out = torch.empty((B, cap), dtype=torch.uint8, device="cuda")
    d_sizes = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_st = torch.ones(B, dtype=torch.int32, device="cuda")
    eng.profile(True)
    eng.encode_device(d_frames, B, w, h, 4, 4, 50, True, d_out, cap, d_sizes, d_st)
    torch.cuda.synchronize()
    stages = eng.profile_read()
    assert "k_tok" in stages and "k_emit_tok" in stages, sorted(stages)
    assert not d_st.cpu().numpy().any()
    sizes = d_sizes.cpu().numpy()
    for f in (0, 1, B // 2, B - 1):
        want = ol.oracle_encode(frames[f], 50, True)
        assert int(sizes[f]) == want.size and np.array_equal(d_out[f, : want.size].cpu().numpy(), want), f
    # the expanded slots of a frame in the middle of the batch
    _, tr = ol.oracle_encode(frames[B // 2], 50, True, trace=True)
    assert np.array_equal(eng.debug_read("fres_tok_sym", B // 2, tr["fres_sym"].size), tr["fres_sym"])
    eng.profile_reset()
    eng.encode_device(d_frames[:1], 1, w, h, 4, 4, 50, True, d_out, cap, d_sizes, d_st)
    torch.cuda.synchronize()
    stages = eng.profile_read()
    assert "k_tok" not in stages and "k_emit" in stages, sorted(stages)
    eng.close()


FRONT_SHAPES = [(512, 64), (1024, 72), (4096, 64), (1920, 136), (200, 72), (2048, 40), (8, 8), (64, 1024), (4096, 1032)]


@pytest.mark.parametrize("w,h", FRONT_SHAPES)
def test_front_kernel_matches_oracle(w, h):
    """k_front (box averages, low-res plane and pixel stage in one pass down the frame, what batches
    take) forced on single frames: every product it leaves -- averages, low-res plane, symbols --
    and the stream against the oracle; shapes with one row, one column of tiles, ragged last
    wavefronts, several chunks of block rows (rows >= 96)."""
    eng = himg_amd.Engine(0)
    eng.set_option("front", 1)
    imgs = [("randtile", himg_amd.synth("randtile", 2, w, h)), ("rand", himg_amd.synth("rand", 5, w, h)),
            ("gradn", himg_amd.synth("gradn", 1, w, h))]
    for name, img in imgs:
        for q, ycc in ((50, True), (90, False), (10, True)):
            want, tr = ol.oracle_encode(img, q, ycc, trace=True)
            got = eng.encode(img, q, ycc)
            n_plane = 4 * tr["rows"] * tr["cols"]
            assert np.array_equal(eng.debug_read("avg", 0, n_plane), tr["avg"]), (name, q, ycc, "box averages")
            assert np.array_equal(eng.debug_read("lowres", 0, n_plane), tr["lowres"]), (name, q, ycc, "low-res plane")
            assert np.array_equal(eng.debug_read("fres_sym", 0, tr["fres_sym"].size), tr["fres_sym"]), (name, q, ycc, "symbols")
            assert got.size == want.size and np.array_equal(got, want), (name, q, ycc)
    eng.close()
