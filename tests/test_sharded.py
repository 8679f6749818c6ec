"""Row-sharded (multi-GPU) encode: partitioning, layout math, the collectives
under gloo with world_size 2 and 3 on CPU, and (GPU) the device phases."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import himg_amd
import oracle_lib as ol
from himg_amd import sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "shard_worker.py")


def test_shard_rows_is_a_16_aligned_partition():
    for rows in (1, 8, 15, 16, 17, 64, 135, 512, 2048):
        for world in (1, 2, 3, 4, 8):
            parts = sharded.shard_rows(rows, world)
            assert parts[0][0] == 0 and parts[-1][1] == rows
            for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(a % 16 == 0 for a, b in parts if b > a)
            # empty shares (more ranks than macro rows) sit at the tail; rank 0 owns rows
            sizes = [b - a for a, b in parts]
            assert sizes[0] > 0 and all(not (x == 0 and y > 0) for x, y in zip(sizes, sizes[1:]))
    assert sharded.shard_rows(512, 8) == [(64 * i, 64 * i + 64) for i in range(8)]


def test_fres_layout_matches_reference_stream():
    img = himg_amd.synth("randtile", 3, 256, 512)
    for q in (10, 90):   # 2-byte headers only / a mix with payloads above 0x7fff? (small rows: all 2-byte)
        stream, tr = ol.oracle_encode(img, q, True, trace=True)
        layout = sharded.fres_layout(tr["fres_row_bytes"].astype(np.int64) * 8, True)
        idx = 12
        while bytes(stream[idx:idx + 4]) != b"FRES":
            idx += 8 + int.from_bytes(bytes(stream[idx + 4:idx + 8]), "little")
        size = int.from_bytes(bytes(stream[idx + 4:idx + 8]), "little")
        base = idx + 8 + tr["fres_tree_bytes"]
        assert layout[3] == size - tr["fres_tree_bytes"]
        for r in range(tr["rows"]):
            n = int(tr["fres_row_bytes"][r])
            h = base + int(layout[0][r])
            assert int(stream[h]) | (int(stream[h + 1]) << 8) == n   # 2-byte header
            assert int(layout[1][r]) - int(layout[0][r]) == 2
    # 4-byte headers above 0x7fff payload bytes
    lay = sharded.fres_layout([8 * 0x7FFF, 8 * 0x8000, 9], True)
    assert list(lay[1] - lay[0]) == [2, 4, 2] and lay[2][2] == 2
    assert sharded.fres_layout([100], False)[3] == 13


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, args, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.parametrize("world,w,h", [(2, 256, 512), (3, 128, 520), (2, 64, 16)])
def test_sharded_orchestration_gloo_cpu(tmp_path, world, w, h):
    """world_size > 1 over gloo on CPU: the device phases are answered by a stub
    built from the oracle, everything else is the product's orchestration."""
    out = tmp_path / "out.himg"
    _run_ranks(world, ["stub", "randtile", 4, w, h, 50, out])
    want = ol.oracle_encode(himg_amd.synth("randtile", 4, w, h), 50, True)
    assert np.array_equal(np.fromfile(out, np.uint8), want)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["assemble", "head"])
@pytest.mark.parametrize("kind,w,h,q,parts", [
    ("randtile", 256, 512, 50, 2), ("randtile", 512, 1024, 50, 4), ("gradn", 256, 264, 90, 3),
    ("rand", 128, 512, 50, 2), ("randtile", 2048, 2048, 50, 8),
    ("randtile", 256, 800, 50, 8), ("gradn", 128, 128, 50, 2)])   # more ranks than macro rows: empty shares
def test_sharded_device_phases_simulated_ranks(kind, w, h, q, parts, form):
    """All ranks simulated in one process (one engine context per rank): the
    exchanges are done by hand exactly as encode_sharded does them -- in the
    final-placement form it runs (rank 0: shard_head, the peers' ranges received at their
    final offsets, shard_finish) and in the gather-then-assemble form of the C ABI's
    multi-device handle."""
    import torch
    img = himg_amd.synth(kind, 1, w, h)
    rows, cols = h // 8, w // 8
    ranges = sharded.shard_rows(rows, parts)
    backs = []
    for (r0, r1) in ranges:
        y0, y1 = max(0, 8 * r0 - 11), min(h, 8 * r1 + 5)
        if r1 <= r0:
            y0, y1 = 0, 1
        d = torch.from_numpy(np.ascontiguousarray(img[y0:y1])).to("cuda:0")
        backs.append(sharded.EngineBackend(himg_amd.Engine(0), d, y0, w, h, q, True))
    stats = [b.stats(r0, r1) for b, (r0, r1) in zip(backs, ranges)]
    hist = sum(s[0] for s in stats)
    bits = [b.row_bits(hist) for b in backs]
    all_bits = torch.cat(bits)
    layout = sharded.fres_layout(all_bits.cpu().numpy(), rows > 1)
    low_full = torch.empty(4 * rows * cols, dtype=torch.uint8, device="cuda:0")
    lf = low_full.view(4, rows, cols)
    for (r0, r1), s in zip(ranges, stats):
        if r1 > r0:
            lf[:, r0:r1, :] = s[1].view(4, r1 - r0, cols)
    if form == "head":
        s0, e0 = sharded.piece_range(layout, *ranges[0])
        buf, base = backs[0].head(low_full, all_bits, s0, e0)
        buf[base + e0: base + layout[3]] = 0xAA            # whatever is not a peer's to send stays rank 0's
        for b, (r0, r1) in list(zip(backs, ranges))[1:]:
            s, e = sharded.piece_range(layout, r0, r1)
            if e > s:
                buf[base + s: base + e] = b.emit(all_bits, s, e)
        out = backs[0].finish(buf)
    else:
        rel_full = torch.empty(layout[3], dtype=torch.uint8, device="cuda:0")
        for b, (r0, r1) in zip(backs, ranges):
            s, e = sharded.piece_range(layout, r0, r1)
            rel_full[s:e] = b.emit(all_bits, s, e)
        out = backs[0].assemble(low_full, all_bits, rel_full)
    want = ol.oracle_encode(img, q, True)
    assert out.size == want.size
    d = np.nonzero(out != want)[0]
    assert d.size == 0, "first mismatch at %d of %d" % (d[0], want.size)
    for b in backs:
        b.eng.close()


@pytest.mark.gpu
def test_sharded_encode_two_processes(tmp_path):
    """Two ranks, two processes, real collectives (gloo with CPU staging because the
    test box has a single GPU), real kernels on both ranks."""
    out = tmp_path / "out.himg"
    _run_ranks(2, ["gpu", "randtile", 2, 512, 1024, 50, out])
    want = ol.oracle_encode(himg_amd.synth("randtile", 2, 512, 1024), 50, True)
    assert np.array_equal(np.fromfile(out, np.uint8), want)


# ---- row-sharded decode ---------------------------------------------------------

@pytest.mark.parametrize("world,kind,w,h,q", [(2, "randtile", 256, 512, 90), (3, "gradn", 128, 520, 50),
                                               (2, "grad", 512, 512, 50), (8, "randtile", 512, 2048, 70)])
def test_sharded_decode_orchestration_gloo_cpu(tmp_path, world, kind, w, h, q):
    """world_size > 1 over gloo on CPU: rank 0 indexes the rows once, broadcasts the
    head of the stream and sends every rank only its own rows' bytes (the stub checks
    what each rank holds), status reduction and the gather of the pixels; the device
    phase is answered by the oracle.  The `grad` case is a stream the reference rejects
    (trap T2).  With 8 ranks: what leaves rank 0 stays below 1.2 x the stream."""
    out = tmp_path / "out.bin"
    _run_ranks(world, ["dstub", kind, 4, w, h, q, out])
    rc, pix = ol.oracle_decode(ol.oracle_encode(himg_amd.synth(kind, 4, w, h), q, True))
    if rc != 0:
        assert open(out).read() == "REJECTED"
    else:
        assert np.array_equal(np.fromfile(out, np.uint8), pix.ravel())
    sent, size = (int(x) for x in open(str(out) + ".stats").read().split())
    assert sent <= 1.2 * size, (sent, size)


def test_index_host_matches_stream_layout():
    """himg_hip_index_host (no GPU): offsets / lengths of the FRES rows and the first
    row header agree with the oracle's per-row byte counts and the chunk layout."""
    img = himg_amd.synth("randtile", 1, 512, 256)
    packed, tr = ol.oracle_encode(img, 90, True, trace=True)
    w, h, c, off, ln, first = himg_amd.index_host(packed)
    assert (w, h, c) == (512, 256, 4)
    assert np.array_equal(ln, tr["fres_row_bytes"])
    lay = sharded.fres_layout(tr["fres_row_bytes"].astype(np.int64) * 8, True)
    assert np.array_equal(off - first, lay[1])
    assert int(off[-1]) + int(ln[-1]) == packed.size
    bad = packed.copy()
    bad[first] ^= 0x40          # the first row's size header now points past the chunk
    bad[first + 1] |= 0x7f
    with pytest.raises(himg_amd.HimgError):
        himg_amd.index_host(bad)
    # slices: the head plus every rank's slice cover each row's payload, nothing else is needed
    parts = sharded.shard_rows(32, 3)
    rng = sharded.slice_ranges(off, ln, first, packed.size, parts)
    for (r0, r1), (lo, hi) in zip(parts, rng):
        if r1 > r0:
            assert lo % 16 == 0 and lo >= first // 16 * 16 and lo <= off[r0] and hi >= off[r1 - 1] + ln[r1 - 1]
        else:
            assert (lo, hi) == (0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,w,h,q,parts", [
    ("randtile", 256, 512, 90, 2), ("randtile", 4096, 256, 50, 2), ("gradn", 256, 261, 90, 3),
    ("randtile", 8192, 128, 50, 2), ("rand", 512, 1024, 50, 4)])
def test_sharded_decode_simulated_ranks(kind, w, h, q, parts):
    """Every rank simulated in one process: each decodes only its block rows
    (fused row kernel, generic path for 8192-wide rows, ragged height) and the
    concatenation equals the oracle's pixels."""
    import torch
    packed = ol.oracle_encode(himg_amd.synth(kind, 2, w, h), q, True)
    rc, pix = ol.oracle_decode(packed)
    assert rc == 0
    pix = pix.reshape(h, w, 4)
    d_packed = torch.zeros((packed.size + 15) // 16 * 16, dtype=torch.uint8, device="cuda:0")
    d_packed[: packed.size] = torch.from_numpy(packed).to("cuda:0")
    rows = (h + 7) // 8
    for (r0, r1) in sharded.shard_rows(rows, parts):
        eng = himg_amd.Engine(0)
        y0, y1 = min(8 * r0, h), min(8 * r1, h)
        d_rows = torch.full((max(y1 - y0, 1), w, 4), 77, dtype=torch.uint8, device="cuda:0")
        st = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        eng.decode_rows_device(d_packed, packed.size, w, h, 4, r0, r1, d_rows, st)
        torch.cuda.synchronize()
        assert int(st.item()) == 0
        if y1 > y0:
            assert np.array_equal(d_rows[: y1 - y0].cpu().numpy(), pix[y0:y1]), (r0, r1)
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,w,h,q,parts", [
    ("randtile", 256, 512, 90, 2), ("randtile", 4096, 256, 50, 2), ("gradn", 256, 261, 90, 3),
    ("randtile", 8192, 128, 50, 2), ("rand", 512, 1024, 50, 4)])
def test_sharded_decode_scattered_stream(kind, w, h, q, parts):
    """The scattered form: the row index from the GPU equals the host's; every simulated
    rank decodes its block rows from a buffer that holds ONLY the head of the stream and
    its own rows' bytes (everything else poisoned), with the index supplied."""
    import torch
    packed = ol.oracle_encode(himg_amd.synth(kind, 2, w, h), q, True)
    rc, pix = ol.oracle_decode(packed)
    assert rc == 0
    pix = pix.reshape(h, w, 4)
    rows = (h + 7) // 8
    _, _, _, off, ln, first = himg_amd.index_host(packed)
    cap = (packed.size + 15) // 16 * 16 + 64
    d_full = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    d_full[: packed.size] = torch.from_numpy(packed).to("cuda:0")
    eng = himg_amd.Engine(0)
    d_idx = torch.zeros(2 * rows + 2, dtype=torch.int32, device="cuda:0")
    st = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    eng.decode_index_device(d_full, packed.size, w, h, 4, d_idx, d_idx[2 * rows:], st)
    got = d_idx.cpu().numpy().view(np.uint32)
    assert int(st.item()) == 0 and int(got[2 * rows]) == first
    assert np.array_equal(got[:rows], off) and np.array_equal(got[rows:2 * rows], ln)
    ranges = sharded.shard_rows(rows, parts)
    for (r0, r1), (lo, hi) in zip(ranges, sharded.slice_ranges(off, ln, first, packed.size, ranges)):
        d_part = torch.full((cap,), 0xAA, dtype=torch.uint8, device="cuda:0")
        head = (first + 15) // 16 * 16
        d_part[:head] = d_full[:head]
        d_part[lo:hi] = d_full[lo:hi]
        y0, y1 = min(8 * r0, h), min(8 * r1, h)
        d_rows = torch.full((max(y1 - y0, 1), w, 4), 77, dtype=torch.uint8, device="cuda:0")
        st.zero_()
        eng.decode_rows_indexed_device(d_part, packed.size, w, h, 4, r0, r1, d_idx, d_rows, st)
        torch.cuda.synchronize()
        assert int(st.item()) == 0
        if y1 > y0:
            assert np.array_equal(d_rows[: y1 - y0].cpu().numpy(), pix[y0:y1]), (r0, r1)
    # a row index that points outside the stream is refused like a damaged header
    bad = d_idx.clone()
    bad[0] = packed.size + 100
    d_rows = torch.zeros((min(8, h), w, 4), dtype=torch.uint8, device="cuda:0")
    eng.decode_rows_indexed_device(d_full, packed.size, w, h, 4, 0, 1, bad, d_rows, st)
    torch.cuda.synchronize()
    assert int(st.item()) != 0
    eng.close()


@pytest.mark.gpu
def test_damaged_rows_from_a_poisoned_scattered_buffer():
    """Streams with a flipped bit in a block row's payload (half of them in the row's last
    bytes, where a token may run into the bytes behind the row), decoded by row ranges from
    buffers that hold only the head and the range's own bytes -- everything else poisoned,
    the tail behind the stream zeroed as ShardedDecoder leaves it: the verdict and the
    pixels are the oracle's, i.e. nothing depends on bytes a rank does not own."""
    import torch
    w, h, q, parts = 512, 256, 70, 2
    packed = ol.oracle_encode(himg_amd.synth("randtile", 3, w, h), q, True)
    rows = (h + 7) // 8
    _, _, _, off, ln, first = himg_amd.index_host(packed)
    cap = (packed.size + 15) // 16 * 16 + 64
    eng = himg_amd.Engine(0)
    ranges = sharded.shard_rows(rows, parts)
    slices = sharded.slice_ranges(off, ln, first, packed.size, ranges)
    d_idx = torch.from_numpy(np.concatenate([off, ln]).astype(np.uint32).view(np.int32)).to("cuda:0")
    st = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    rng = np.random.default_rng(11)
    accepted = rejected = 0
    for trial in range(32):
        m = packed.copy()
        r = int(rng.integers(rows))
        if trial % 2:
            pos = int(off[r]) + int(ln[r]) - 1 - int(rng.integers(min(3, int(ln[r]))))
        else:
            pos = int(off[r]) + int(rng.integers(int(ln[r])))
        m[pos] ^= np.uint8(1 << int(rng.integers(8)))
        rc, pix = ol.oracle_decode(m)
        d_full = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
        d_full[: m.size] = torch.from_numpy(m).to("cuda:0")
        bad = False
        got = np.zeros((h, w, 4), np.uint8)
        for (r0, r1), (lo, hi) in zip(ranges, slices):
            d_part = torch.full((cap,), 0xAA, dtype=torch.uint8, device="cuda:0")
            head = (first + 15) // 16 * 16
            d_part[:head] = d_full[:head]
            d_part[lo:hi] = d_full[lo:hi]
            d_part[m.size:] = 0
            y0, y1 = min(8 * r0, h), min(8 * r1, h)
            d_rows = torch.zeros((max(y1 - y0, 1), w, 4), dtype=torch.uint8, device="cuda:0")
            st.zero_()
            eng.decode_rows_indexed_device(d_part, m.size, w, h, 4, r0, r1, d_idx, d_rows, st)
            torch.cuda.synchronize()
            bad = bad or int(st.item()) != 0
            got[y0:y1] = d_rows[: y1 - y0].cpu().numpy()
        assert bad == (rc != 0), (trial, r, pos)
        if rc == 0:
            assert np.array_equal(got, pix.reshape(h, w, 4)), (trial, r, pos)
            accepted += 1
        else:
            rejected += 1
    assert accepted and rejected   # both outcomes occur among the mutations
    eng.close()


@pytest.mark.gpu
def test_sharded_decode_two_processes(tmp_path):
    """Two ranks, two processes, real collectives (gloo with CPU staging: one GPU on
    the test box), real kernels on both ranks; and a stream both ranks reject."""
    out = tmp_path / "out.bin"
    _run_ranks(2, ["dgpu", "randtile", 2, 512, 1024, 70, out])
    rc, pix = ol.oracle_decode(ol.oracle_encode(himg_amd.synth("randtile", 2, 512, 1024), 70, True))
    assert rc == 0 and np.array_equal(np.fromfile(out, np.uint8), pix.ravel())
    _run_ranks(2, ["dgpu", "grad", 0, 512, 512, 50, out])
    assert open(out).read() == "REJECTED"


@pytest.mark.gpu
def test_sharded_decode_stream_in_rank0_hbm(tmp_path):
    """The stream lives in rank 0's HBM: rank 0 learns where the first row header lies
    (himg_hip_decode_first_device), the head goes out, the header walk runs on its GPU's side
    stream BESIDE the head phase (himg_hip_decode_walk_device), then the rows' bytes travel --
    same pixels, same verdicts (a damaged row header; a stream the reference rejects)."""
    out = tmp_path / "out.bin"
    _run_ranks(2, ["dgpu_dev", "randtile", 2, 512, 1024, 70, out])
    rc, pix = ol.oracle_decode(ol.oracle_encode(himg_amd.synth("randtile", 2, 512, 1024), 70, True))
    assert rc == 0 and np.array_equal(np.fromfile(out, np.uint8), pix.ravel())
    _run_ranks(3, ["dgpu_dev", "gradn", 5, 256, 520, 90, out])
    rc, pix = ol.oracle_decode(ol.oracle_encode(himg_amd.synth("gradn", 5, 256, 520), 90, True))
    assert rc == 0 and np.array_equal(np.fromfile(out, np.uint8), pix.ravel())
    _run_ranks(2, ["dgpu_dev", "grad", 0, 512, 512, 50, out])
    assert open(out).read() == "REJECTED"


def _gpus():
    import torch
    return torch.cuda.device_count()   # (does not initialise the GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sharded_rccl_one_gpu_per_rank(tmp_path, world):
    """The row-sharded encode and decode over RCCL (backend "nccl"), one GPU per rank,
    device tensors in every collective and point-to-point transfer -- the path
    `bench.py --gpus N` takes on a multi-GPU node.  Skipped where fewer GPUs are visible
    (the single-GPU test box: the gloo tests above run the same orchestration there)."""
    if _gpus() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, _gpus()))
    out = tmp_path / "out.himg"
    img = himg_amd.synth("randtile", 3, 1024, 2048)
    want = ol.oracle_encode(img, 50, True)
    _run_ranks(world, ["gpu_nccl", "randtile", 3, 1024, 2048, 50, out])
    assert np.array_equal(np.fromfile(out, np.uint8), want)
    rc, pix = ol.oracle_decode(want)
    assert rc == 0
    _run_ranks(world, ["dgpu_nccl", "randtile", 3, 1024, 2048, 50, out])
    assert np.array_equal(np.fromfile(out, np.uint8), pix.ravel())
