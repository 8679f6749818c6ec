"""The two BASELINE configurations the per-shape parity tests do not reach at full
size, against golden tables recorded from the REAL reference:

  config 3  256 independent 1920x1080 RGBA frames (randtile seeds 0..255, q=50) in
            ONE himg_hip_encode_device launch, every stream checked; then decoded in
            one launch, every frame checked  (tests/golden/batch_1920x1080_q50.json,
            made by tests/golden/make_golden_batch.py)
  config 4  one 16384x16384 RGBA frame, block rows sharded over 8 ranks
            (himg_hip_shard_* phases, the exchanges done by hand exactly as
            himg_amd/sharded.py does them over RCCL), then decoded by the same 8 row
            ranges (himg_hip_decode_rows_device)  (tests/golden/golden.json)
"""
import json
import os

import numpy as np
import pytest

import himg_amd
from golden_util import GOLDEN, GOLDEN_DIR
from himg_amd import sharded

pytestmark = pytest.mark.gpu


def _batch_table(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def test_config3_256_frames_1080p_one_launch(engine):
    import torch
    tab = _batch_table("batch_1920x1080_q50.json")
    W, H, Q, B = tab["width"], tab["height"], tab["quality"], 256
    assert len(tab["seeds"]) == B
    frames = np.stack([himg_amd.synth(tab["kind"], s, W, H) for s in range(B)])
    d_frames = torch.from_numpy(frames).to("cuda:0")
    cap = himg_amd.max_packed_size(W, H, 4)
    d_out = torch.empty((B, cap), dtype=torch.uint8, device="cuda:0")
    d_sizes = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    d_st = torch.ones(B, dtype=torch.int32, device="cuda:0")
    engine.encode_device(d_frames, B, W, H, 4, 4, Q, True, d_out, cap, d_sizes, d_st)
    torch.cuda.synchronize()
    assert not d_st.cpu().numpy().any()
    sizes = d_sizes.cpu().numpy().astype(np.uint32)
    out = d_out.cpu().numpy()
    for s in range(B):
        want_size, want_fnv, _ = tab["seeds"][s]
        assert int(sizes[s]) == want_size, (s, int(sizes[s]), want_size)
        assert himg_amd.fnv1a64(out[s, :want_size]) == want_fnv, "stream of seed %d differs from the reference" % s
    # ... and back: one decode launch over the 256 streams.
    d_pix = torch.empty((B, H, W, 4), dtype=torch.uint8, device="cuda:0")
    d_st2 = torch.ones(B, dtype=torch.int32, device="cuda:0")
    engine.decode_device(d_out, cap, sizes, B, W, H, 4, d_pix, d_st2)
    torch.cuda.synchronize()
    assert not d_st2.cpu().numpy().any()
    pix = d_pix.cpu().numpy()
    for s in range(B):
        assert himg_amd.fnv1a64(pix[s]) == tab["seeds"][s][2], "pixels of seed %d differ from the reference" % s


def test_config4_16384_rows_sharded_over_8_ranks():
    import torch
    rec = GOLDEN["randtile_s0_16384x16384_q50"]
    W = H = 16384
    Q, parts = 50, 8
    dev = "cuda:0"
    img = himg_amd.synth("randtile", 0, W, H)
    assert himg_amd.fnv1a64(img) == rec["input_fnv"]
    rows, cols = H // 8, W // 8
    ranges = sharded.shard_rows(rows, parts)
    backs = []
    for (r0, r1) in ranges:
        y0, y1 = max(0, 8 * r0 - 11), min(H, 8 * r1 + 5)   # the shard plus its halo
        d = torch.from_numpy(np.ascontiguousarray(img[y0:y1])).to(dev)
        backs.append(sharded.EngineBackend(himg_amd.Engine(0), d, y0, W, H, Q, True))
    del img
    stats = [b.stats(r0, r1) for b, (r0, r1) in zip(backs, ranges)]
    hist = sum(s[0] for s in stats)                           # = all-reduce(sum)
    bits = [b.row_bits(hist) for b in backs]
    all_bits = torch.cat(bits)                                # = all-gather
    layout = sharded.fres_layout(all_bits.cpu().numpy(), True)
    low_full = torch.empty(4 * rows * cols, dtype=torch.uint8, device=dev)
    lf = low_full.view(4, rows, cols)
    for (r0, r1), s in zip(ranges, stats):                    # = gather of the low-res rows
        lf[:, r0:r1, :] = s[1].view(4, r1 - r0, cols)
    # Rank 0 in its final-placement form (what encode_sharded runs): LRES stream, container,
    # tree, all row headers and its own rows straight into the stream buffer; the peers'
    # byte ranges land at their final offsets (the receive of sharded.py, by hand), then
    # the pad bits.
    s0, e0 = sharded.piece_range(layout, *ranges[0])
    buf, base = backs[0].head(low_full, all_bits, s0, e0)
    for b, (r0, r1) in list(zip(backs, ranges))[1:]:          # = the peers' sends, received in place
        s, e = sharded.piece_range(layout, r0, r1)
        buf[base + s: base + e] = b.emit(all_bits, s, e)
    stream = backs[0].finish(buf, host=False)
    for b in backs:
        b.eng.close()
    del backs, stats
    assert stream.numel() == rec["packed_size"] == 275620945
    assert himg_amd.fnv1a64(stream.cpu().numpy()) == rec["stream_fnv"] == "5bdcdb7a140df481"

    # Decode by the same 8 row ranges (pixels stay sharded on a real node; here they
    # land in one image so that it can be hashed).
    size = stream.numel()
    d_packed = torch.zeros((size + 15) // 16 * 16, dtype=torch.uint8, device=dev)
    d_packed[:size] = stream
    del stream
    d_img = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    for (r0, r1) in ranges:
        eng = himg_amd.Engine(0)
        st = torch.ones(1, dtype=torch.int32, device=dev)
        eng.decode_rows_device(d_packed, size, W, H, 4, r0, r1, d_img[8 * r0: 8 * r1], st)
        torch.cuda.synchronize()
        assert int(st.item()) == 0
        eng.close()
    assert himg_amd.fnv1a64(d_img.cpu().numpy()) == rec["decoded_fnv"] == "08fb9dc8e25c2fae"
