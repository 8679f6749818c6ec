"""Row-sharded encode of ONE large frame over several GPUs (BASELINE config 4).

FRES block rows are independently coded units behind size headers (reference
huffman_enc.cpp:342-358), so the frame shards by block rows, aligned to 16 rows
(one low-res macro-block row).  One process per GPU; this module runs the
collectives of SURVEY.md 8(e) between the device phases of the C ABI
(include/himg_hip.h, himg_hip_shard_*):

    stats      -> all-reduce(sum) 261-bin FRES token histogram   (1 KiB, latency bound)
               -> gather low-res rows to rank 0                  (1/64 of the pixels)
    row_bits   -> all-gather payload bits of every block row     (rows x u32)
    emit       -> gather the packed rows to rank 0               (the only large message;
                                                                   each peer -> rank 0 over
                                                                   its own xGMI link)
    assemble   (rank 0) LRES stream, container, FRES tree, rows, stale pad bits

The result on rank 0 is byte-identical to encoding the whole frame on one GPU.
`backend` hides the device: EngineBackend drives the HIP engine; the tests use a
stub to exercise exactly this orchestration under gloo on CPU.
"""
import numpy as np


def shard_rows(rows, world):
    """Split `rows` block rows over `world` ranks in multiples of 16 rows (a
    low-res macro-block row never straddles two ranks).  The macro rows are dealt
    out as evenly as possible with the larger shares first, so ranks that get
    nothing (more ranks than macro rows) are at the TAIL and rank 0, which
    assembles the stream, always owns rows."""
    macro = (rows + 15) // 16
    base, rem = divmod(macro, world)
    out, m0 = [], 0
    for r in range(world):
        m1 = m0 + base + (1 if r < rem else 0)
        out.append((min(16 * m0, rows), min(16 * m1, rows)))
        m0 = m1
    return out


def fres_layout(row_bits, use_blocks):
    """Byte layout of the FRES rows relative to the first row header
    (huffman_enc.cpp:342-358): per row a 2-byte size header, or 4 bytes when the
    payload exceeds 0x7fff bytes; none for single-row images.
    Returns (header_off, payload_off, nbytes, total)."""
    bits = np.asarray(row_bits, np.int64)
    nbytes = (bits + 7) >> 3
    hdr = np.where(nbytes <= 0x7FFF, 2, 4) if use_blocks else np.zeros_like(nbytes)
    step = hdr + nbytes
    header_off = np.concatenate([[0], np.cumsum(step)[:-1]]) if len(step) else np.zeros(0, np.int64)
    return header_off, header_off + hdr, nbytes, int(step.sum())


def piece_range(layout, r0, r1):
    """Byte range [start, end) of rows [r0, r1) in the relative FRES layout."""
    header_off, payload_off, nbytes, total = layout
    if r0 >= r1:
        return 0, 0
    return int(header_off[r0]), int(payload_off[r1 - 1] + nbytes[r1 - 1])


def _pad(t, n):
    import torch
    if t.numel() == n:
        return t.contiguous()
    out = torch.zeros(n, dtype=t.dtype, device=t.device)
    out[: t.numel()] = t.reshape(-1)
    return out


def encode_sharded(backend, rows, cols, channels, use_blocks, group=None, host=True):
    """Run the sharded encode.  Returns the packed stream on rank 0 (a uint8 numpy
    array; with host=False whatever backend.assemble leaves where it was built --
    the engine backend returns a CUDA tensor, sparing the 268 MB D2H copy of a
    16384x16384 stream) and None elsewhere.  `backend` provides:
        stats(r0, r1)          -> (hist int64[261], low uint8[C*(r1-r0)*cols])   comm tensors
        row_bits(hist_global)  -> int32[r1-r0]
        emit(all_bits int32[rows], start, end) -> uint8[end-start]   (the local byte range)
        assemble(low_full uint8[C*rows*cols], all_bits, rel_full uint8[total], host) -> numpy uint8 / tensor
    All tensors live where the process group can move them (CUDA for nccl/RCCL,
    CPU for gloo)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    parts = shard_rows(rows, world)
    r0, r1 = parts[rank]
    max_rows = max(b - a for a, b in parts)

    hist, low = backend.stats(r0, r1)
    if world > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)        # 261 x i64, latency bound
    # Low-res rows to rank 0 (1/64 of the pixels): padded gather.
    low_pad = _pad(low, channels * max_rows * cols)
    if world > 1:
        low_list = [torch.empty_like(low_pad) for _ in range(world)] if rank == 0 else None
        dist.gather(low_pad, low_list, dst=0, group=group)
    else:
        low_list = [low_pad]

    bits = backend.row_bits(hist)
    bits_pad = _pad(bits, max_rows)
    if world > 1:
        bits_list = [torch.empty_like(bits_pad) for _ in range(world)]
        dist.all_gather(bits_list, bits_pad, group=group)                 # rows x i32
    else:
        bits_list = [bits_pad]
    all_bits = torch.cat([b[: (p1 - p0)] for b, (p0, p1) in zip(bits_list, parts)])
    layout = fres_layout(all_bits.cpu().numpy(), use_blocks)
    ranges = [piece_range(layout, p0, p1) for p0, p1 in parts]
    max_piece = max(e - s for s, e in ranges)

    start, end = ranges[rank]
    piece = backend.emit(all_bits, start, end)
    piece_pad = _pad(piece, max_piece)
    if world > 1:
        piece_list = [torch.empty_like(piece_pad) for _ in range(world)] if rank == 0 else None
        dist.gather(piece_pad, piece_list, dst=0, group=group)           # the large message
    else:
        piece_list = [piece_pad]
    if rank != 0:
        return None

    total = layout[3]
    if world == 1:
        # One rank holds everything already: no copies (a 16384x16384 frame's packed rows
        # are 275 MB).
        return backend.assemble(low, all_bits, piece, host)
    rel_full = torch.empty(total, dtype=torch.uint8, device=piece.device)
    for pl, (s, e) in zip(piece_list, ranges):
        rel_full[s:e] = pl[: e - s]
    low_full = torch.empty(channels * rows * cols, dtype=torch.uint8, device=low.device)
    lf = low_full.view(channels, rows, cols)
    for ll, (p0, p1) in zip(low_list, parts):
        if p1 > p0:
            lf[:, p0:p1, :] = ll[: channels * (p1 - p0) * cols].view(channels, p1 - p0, cols)
    return backend.assemble(low_full, all_bits, rel_full, host)


class EngineBackend:
    """Device phases on the HIP engine.  `d_frame` is this rank's view of the
    frame: a CUDA uint8 tensor holding pixel rows [y_first, y_first + n) of the
    W x H image (at least rows 8*r0-11 .. 8*r1+4 clipped to the image).

    Every buffer the phases touch is allocated once, here; the phases only launch
    kernels on the (torch current = null) stream, so the collectives that follow
    them are ordered behind them without a host synchronisation.  The one value the
    host needs per frame is the row layout (rows x 4 bytes, for the sizes of the
    pieces that travel to rank 0)."""

    def __init__(self, engine, d_frame, y_first, width, height, quality=50, use_ycbcr=True,
                 comm_device=None, stream=0):
        import torch
        import himg_amd
        self.eng, self.W, self.H, self.q, self.ycbcr = engine, width, height, quality, use_ycbcr
        self.d_frame, self.y_first, self.stream = d_frame, y_first, stream
        self.dev = d_frame.device
        self.comm = torch.device(comm_device) if comm_device is not None else self.dev
        self.rows, self.cols, self.C = (height + 7) // 8, (width + 7) // 8, 4
        self.r0 = self.r1 = 0
        dev = self.dev
        self._hist32 = torch.zeros(264, dtype=torch.int32, device=dev)
        self._hist_g = torch.zeros(264, dtype=torch.int32, device=dev)
        self._low = None        # sized at the first stats() (depends on the share)
        self._bits = None
        self._rel_cap = (self.rows * self.cols * 64 * self.C + 4 * self.rows + 256 + 255) // 256 * 256
        self._rel = None        # allocated on first emit: only as large as the local rows can get
        self._size = torch.zeros(4, dtype=torch.int32, device=dev)
        self._status = torch.zeros(4, dtype=torch.int32, device=dev)
        self._out = None        # rank 0 only (assemble)
        self._out_cap = himg_amd.max_packed_size(width, height, self.C)

    def _to_comm(self, t):
        return t.to(self.comm) if t.device != self.comm else t

    def stats(self, r0, r1):
        import torch
        if self._low is None or (r0, r1) != (self.r0, self.r1):
            self._low = torch.zeros(max(1, self.C * (r1 - r0) * self.cols), dtype=torch.uint8, device=self.dev)
            self._bits = torch.zeros(max(1, r1 - r0), dtype=torch.int32, device=self.dev)
        self.r0, self.r1 = r0, r1
        # Called for an empty share too: it sets up the per-frame state every later
        # phase (and rank 0's assemble) relies on.
        base = self.d_frame.data_ptr() - self.y_first * self.W * 4   # virtual frame base
        self.eng.shard_stats(base, self.W, self.H, 4, 4, self.q, self.ycbcr, r0, r1, self._hist32, self._low,
                             self.stream)
        h64 = (self._hist32[:261].to(torch.int64) & 0xFFFFFFFF)    # i64: 8 ranks x 2^31 tokens cannot wrap
        return self._to_comm(h64), self._to_comm(self._low[: self.C * (r1 - r0) * self.cols])

    def row_bits(self, hist_global):
        import torch
        n = self.r1 - self.r0
        self._hist_g[:261] = hist_global.to(self.dev).to(torch.int32)
        self.eng.shard_row_bits(self._hist_g, self._bits, self.stream)   # every rank builds the (identical) tree
        return self._to_comm(self._bits[:n])

    def emit(self, all_bits, start, end):
        import torch
        if end <= start:
            return torch.zeros(0, dtype=torch.uint8, device=self.comm)
        if self._rel is None:
            self._rel = torch.empty(self._rel_cap, dtype=torch.uint8, device=self.dev)
        self._all_bits_dev = all_bits.to(self.dev).to(torch.int32).contiguous()
        self.eng.shard_emit(self._all_bits_dev, self._rel, self._rel_cap, self._size, self.stream)
        return self._to_comm(self._rel[start:end])

    def assemble(self, low_full, all_bits, rel_full, host=True):
        import torch
        import himg_amd
        if self._out is None:
            self._out = torch.empty(self._out_cap, dtype=torch.uint8, device=self.dev)
        bits_dev = all_bits.to(self.dev).to(torch.int32).contiguous()
        rel_dev = rel_full.to(self.dev).contiguous()
        self.eng.shard_assemble(low_full.to(self.dev).contiguous(), bits_dev, rel_dev, rel_dev.numel(),
                                self._out, self._out_cap, self._size, self._status, self.stream)
        st = torch.stack([self._size[0], self._status[0]]).cpu()       # the one wait of this phase
        if int(st[1]) != 0:
            raise himg_amd.HimgError(-int(st[1]), "sharded assemble failed")
        out = self._out[: int(st[0])]
        return out.cpu().numpy() if host else out


# ---------------------------------------------------------------------------
# Row-sharded DECODE of one frame.
# ---------------------------------------------------------------------------

def decode_sharded(engine, packed, width, height, channels=4, group=None, gather=True, device=None,
                   comm_device=None, stream=0):
    """Decode one frame with its block rows sharded over the ranks of `group`.

    `packed` (uint8 numpy array or tensor) is the whole stream on rank 0; it is
    broadcast (the only exchange before the kernels -- the stream is ~4x smaller
    than the pixels).  Every rank decodes the LRES stream and its own block rows
    (reference decoder.cpp:292-326 hands block rows to worker threads the same
    way).  Returns (ok, pixels): `ok` is the AND over the ranks (a stream the
    reference rejects is rejected); with gather=True rank 0 gets the whole
    H x W x C image (None elsewhere), otherwise every rank gets its own pixel rows.
    `engine` provides decode_rows_device (himg_amd.Engine); `comm_device` is where
    the process group can move tensors (CUDA for nccl/RCCL, "cpu" for gloo).
    """
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    comm = torch.device(comm_device) if comm_device is not None else dev
    rows = (height + 7) // 8
    parts = shard_rows(rows, world)
    r0, r1 = parts[rank]

    def bcast(t):
        if world > 1:
            c = t.to(comm)
            dist.broadcast(c, src=0, group=group)
            if c is not t:
                t.copy_(c)

    # The stream, padded to a multiple of 16 bytes, on every rank.
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        n[0] = int(packed.numel() if torch.is_tensor(packed) else len(packed))
    bcast(n)
    size = int(n.item())
    d_packed = torch.zeros((size + 15) // 16 * 16, dtype=torch.uint8, device=dev)
    if rank == 0:
        src = packed if torch.is_tensor(packed) else torch.from_numpy(np.ascontiguousarray(packed, np.uint8))
        d_packed[:size] = src.to(dev)
    bcast(d_packed)

    y0, y1 = min(8 * r0, height), min(8 * r1, height)
    d_rows = torch.empty((max(y1 - y0, 1), width, channels), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    engine.decode_rows_device(d_packed, size, width, height, channels, r0, r1, d_rows, d_status, stream)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    bad = (d_status != 0).to(torch.int32).to(comm)
    if world > 1:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    ok = int(bad.item()) == 0
    d_rows = d_rows[: y1 - y0]
    if not gather:
        return ok, (d_rows if ok else None)
    if world == 1:
        return ok, (d_rows.cpu().numpy() if ok else None)
    # Gather the pixel rows on rank 0 (padded to the largest shard).
    max_rows = max(min(8 * b, height) - min(8 * a, height) for a, b in parts)
    pad = torch.zeros((max_rows, width, channels), dtype=torch.uint8, device=comm)
    pad[: y1 - y0] = d_rows.to(comm)
    lst = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, lst, dst=0, group=group)
    if rank != 0 or not ok:
        return ok, None
    out = torch.empty((height, width, channels), dtype=torch.uint8, device=comm)
    for t, (a, b) in zip(lst, parts):
        ya, yb = min(8 * a, height), min(8 * b, height)
        out[ya:yb] = t[: yb - ya]
    return ok, out.cpu().numpy()
