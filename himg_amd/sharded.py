"""Row-sharded encode of ONE large frame over several GPUs (BASELINE config 4).

FRES block rows are independently coded units behind size headers (reference
huffman_enc.cpp:342-358), so the frame shards by block rows, aligned to 16 rows
(one low-res macro-block row).  One process per GPU; this module runs the
collectives of SURVEY.md 8(e) between the device phases of the C ABI
(include/himg_hip.h, himg_hip_shard_*):

    stats      -> all-reduce(sum) 261-bin FRES token histogram   (1 KiB, latency bound)
               -> low-res rows to rank 0, point to point         (1/64 of the pixels)
    row_bits   -> all-gather payload bits of every block row     (rows x u32)
    emit       -> the packed rows to rank 0, point to point,     (the only large message;
                  exact sizes, received at their FINAL offsets    each peer -> rank 0 over
                  in rank 0's stream buffer                       its own xGMI link)
    head       (rank 0, WHILE the peers pack and send) LRES stream, container, FRES tree,
               every row's size header, its own rows -- straight into the stream buffer
    finish     (rank 0) stale pad bits, once every row has arrived

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
      or, final-placement form (used when the backend has it):
        head(low_full, all_bits, own_start, own_end) -> (stream buffer uint8[>= size], offset of the first row header)
        finish(buffer, host) -> numpy uint8 / tensor
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
    # Low-res rows to rank 0 (1/64 of the pixels), exact sizes, point to point.
    low_list = [low]
    if world > 1:
        ops = []
        if rank == 0:
            low_list += [torch.empty(channels * (p1 - p0) * cols, dtype=torch.uint8, device=low.device)
                         for p0, p1 in parts[1:]]
            ops = [dist.P2POp(dist.irecv, t, peer, group) for peer, t in enumerate(low_list) if peer and t.numel()]
        elif low.numel():
            ops = [dist.P2POp(dist.isend, low.contiguous(), 0, group)]
        low_reqs = dist.batch_isend_irecv(ops) if ops else []
    else:
        low_reqs = []

    bits = backend.row_bits(hist)
    bits_pad = _pad(bits, max_rows)
    if world > 1:
        bits_list = [torch.empty_like(bits_pad) for _ in range(world)]
        dist.all_gather(bits_list, bits_pad, group=group)                 # rows x i32
    else:
        bits_list = [bits_pad]
    all_bits = torch.cat([b[: (p1 - p0)] for b, (p0, p1) in zip(bits_list, parts)])
    # The one value the host needs: the sizes of the pieces that travel (rows x 4 bytes).
    layout = fres_layout(all_bits.cpu().numpy(), use_blocks)
    ranges = [piece_range(layout, p0, p1) for p0, p1 in parts]
    total = layout[3]

    start, end = ranges[rank]
    if hasattr(backend, "head"):
        # Final-placement form.  Rank 0 does not pack into a relative buffer and assemble
        # afterwards: as soon as the row sizes are known it builds everything that does not
        # come from a peer -- LRES stream, container, tree, every row header, its own rows --
        # in the stream buffer itself, learns where the first row header lies (one 8-byte
        # read-back), and receives every peer's byte range AT ITS FINAL OFFSET, each over
        # that peer's own link.  Its LRES branch and its own packing run while the peers
        # pack and send; behind the last receive only the pad-bit fix-up is left.
        if rank != 0:
            piece = backend.emit(all_bits, start, end)
            reqs = list(low_reqs)
            if end > start:
                reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, piece.contiguous(), 0, group)])
            for req in reqs:
                req.wait()
            return None
        for req in low_reqs:
            req.wait()
        if world > 1:
            low_full = torch.empty(channels * rows * cols, dtype=torch.uint8, device=low.device)
            lf = low_full.view(channels, rows, cols)
            for ll, (p0, p1) in zip(low_list, parts):
                if p1 > p0:
                    lf[:, p0:p1, :] = ll[: channels * (p1 - p0) * cols].view(channels, p1 - p0, cols)
        else:
            low_full = low
        out, base = backend.head(low_full, all_bits, start, end)   # `out`: the stream buffer as the process group sees it
        ops = [dist.P2POp(dist.irecv, out[base + s: base + e], peer, group)
               for peer, (s, e) in enumerate(ranges) if peer and e > s]
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        return backend.finish(out, host)
    piece = backend.emit(all_bits, start, end)
    if world == 1:
        # One rank holds everything already: no copies (a 16384x16384 frame's packed rows
        # are 275 MB).
        return backend.assemble(low, all_bits, piece, host)
    # The packed rows -- the only large message -- go straight to their place in rank
    # 0's relative FRES buffer: exact sizes, every peer over its own link, no staging
    # copy on either side.
    ops, rel_full = [], None
    if rank == 0:
        rel_full = torch.empty(total, dtype=torch.uint8, device=piece.device)
        rel_full[start:end] = piece
        ops = [dist.P2POp(dist.irecv, rel_full[s:e], peer, group)
               for peer, (s, e) in enumerate(ranges) if peer and e > s]
    elif end > start:
        ops = [dist.P2POp(dist.isend, piece.contiguous(), 0, group)]
    for req in list(low_reqs) + (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
    if rank != 0:
        return None
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
                 comm_device=None, stream=None):
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

    def _s(self):
        """The stream the engine's kernels go to: torch's CURRENT stream of the device
        unless one was given -- so that the torch copies and the collectives around the
        phases, which torch issues on that stream, are ordered with them whatever stream
        context the caller is in."""
        if self.stream is not None:
            return self.stream
        import torch
        return torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else 0

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
                             self._s())
        h64 = (self._hist32[:261].to(torch.int64) & 0xFFFFFFFF)    # i64: 8 ranks x 2^31 tokens cannot wrap
        return self._to_comm(h64), self._to_comm(self._low[: self.C * (r1 - r0) * self.cols])

    def row_bits(self, hist_global):
        import torch
        n = self.r1 - self.r0
        self._hist_g[:261] = hist_global.to(self.dev).to(torch.int32)
        self.eng.shard_row_bits(self._hist_g, self._bits, self._s())   # every rank builds the (identical) tree
        return self._to_comm(self._bits[:n])

    def emit(self, all_bits, start, end):
        import torch
        if end <= start:
            return torch.zeros(0, dtype=torch.uint8, device=self.comm)
        if self._rel is None:
            self._rel = torch.empty(self._rel_cap, dtype=torch.uint8, device=self.dev)
        self._all_bits_dev = all_bits.to(self.dev).to(torch.int32).contiguous()
        self.eng.shard_emit(self._all_bits_dev, self._rel, self._rel_cap, self._size, self._s())
        return self._to_comm(self._rel[start:end])

    def head(self, low_full, all_bits, own_start, own_end):
        """Rank 0, final-placement form: everything but the peers' rows into the stream buffer
        ([own_start, own_end): the byte range of this rank's own rows behind the first row
        header).  Returns (buffer as the process group sees it, offset of the first row header)."""
        self._own_range = (own_start, own_end)
        import torch
        import himg_amd
        if self._out is None:
            self._out = torch.empty(self._out_cap, dtype=torch.uint8, device=self.dev)
            self._head = torch.zeros(4, dtype=torch.int32, device=self.dev)
        bits_dev = all_bits.to(self.dev).to(torch.int32).contiguous()
        self.eng.shard_head(low_full.to(self.dev).contiguous(), bits_dev, self._out, self._out_cap, self._size,
                            self._head, self._status, self._s())
        h = torch.cat([self._head[:2], self._status[:1]]).cpu()      # the one wait of this phase (12 bytes)
        if int(h[2]) != 0 or int(h[1]) == 0:
            raise himg_amd.HimgError(-int(h[2]) if int(h[2]) else himg_amd.HIMG_ERR_UNSUPPORTED, "sharded head failed")
        self._total = int(h[1])
        if self.comm == self.dev:
            return self._out, int(h[0])
        self._stage = torch.empty(self._total, dtype=torch.uint8, device=self.comm)   # (tests: gloo with CPU staging)
        return self._stage, int(h[0])

    def finish(self, out, host=True):
        import torch
        if self.comm != self.dev:
            # The peers' ranges arrived in the staging tensor: move what lies behind the
            # first row header, except this rank's own rows, which head() packed in place.
            s0, e0 = self._own_range
            base = int(self._head[0].item())
            n = self._total
            if s0 > 0:
                self._out[base:base + s0] = out[base:base + s0].to(self.dev)
            if base + e0 < n:
                self._out[base + e0:n] = out[base + e0:n].to(self.dev)
        self.eng.shard_finish(self._out, self._out_cap, self._size, self._s())
        res = self._out[: self._total]
        return res.cpu().numpy() if host else res

    def assemble(self, low_full, all_bits, rel_full, host=True):
        import torch
        import himg_amd
        if self._out is None:
            self._out = torch.empty(self._out_cap, dtype=torch.uint8, device=self.dev)
        bits_dev = all_bits.to(self.dev).to(torch.int32).contiguous()
        rel_dev = rel_full.to(self.dev).contiguous()
        self.eng.shard_assemble(low_full.to(self.dev).contiguous(), bits_dev, rel_dev, rel_dev.numel(),
                                self._out, self._out_cap, self._size, self._status, self._s())
        st = torch.stack([self._size[0], self._status[0]]).cpu()       # the one wait of this phase
        if int(st[1]) != 0:
            raise himg_amd.HimgError(-int(st[1]), "sharded assemble failed")
        out = self._out[: int(st[0])]
        # host=False: a VIEW of this backend's output buffer -- the next encode on the
        # same backend overwrites it; clone it to keep it.
        return out.cpu().numpy() if host else out


# ---------------------------------------------------------------------------
# Row-sharded DECODE of one frame.
# ---------------------------------------------------------------------------

def slice_ranges(offsets, lengths, rows_first, size, parts, margin=16):
    """Byte range [lo, hi) of the stream that the rank decoding block rows [r0, r1)
    needs besides the head [0, rows_first): its rows' payloads (the row kernels read
    whole dwords and a few dwords ahead, hence the margin), 16-byte aligned, clipped to
    [rows_first & ~15, size rounded up to 16).  Empty shares get (0, 0)."""
    cap = (int(size) + 15) // 16 * 16
    out = []
    for r0, r1 in parts:
        if r1 <= r0:
            out.append((0, 0))
            continue
        lo = max(int(offsets[r0]) - margin, int(rows_first)) // 16 * 16
        hi = min((int(offsets[r1 - 1]) + int(lengths[r1 - 1]) + margin + 15) // 16 * 16, cap)
        out.append((lo, max(hi, lo)))
    return out


def _readable_bytes(t):
    """Bytes of `t`'s storage from its first element on (the engine reads the stream in whole
    dwords: a view of a padded buffer can be decoded in place, an exact-size tensor cannot)."""
    try:
        return t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()
    except (AttributeError, RuntimeError):
        return t.numel() * t.element_size()


class ShardedDecoder:
    """Row-sharded decode of frames of one geometry (reference
    decoder.cpp:292-326 hands block rows to worker threads the same way).

    Rank 0 holds the stream.  The exchange is a pipeline of two stages:
        broadcast   a small record: stream size, verdict so far, first-row offset
                    (container chunk look-ups only: no header walk yet)
        broadcast   the head of the stream [0, first row header): container chunks,
                    LRES stream, FRES tree -- 1/20 of a 16384 x 16384 stream
      -> EVERY rank starts what needs only the head (himg_hip_decode_head_device:
         container parse, LRES chain, predictor inverse -- 1 ms of a 16384 x 16384 decode)
         WHILE rank 0 indexes the block rows once (the serial walk over the row size
         headers: microseconds on the host for a stream in host memory, 1.3 ms of
         dependent loads on its GPU otherwise)
        send/recv   to every other rank the index SLICE and ONLY the bytes of its own block
                    rows, each over that peer's own link, the moment the walk has passed them
                    (the walk runs in the ranks' row ranges; a stream in host memory is indexed
                    at once)
      -> every rank decodes its rows (himg_hip_decode_rows_after_head_device) from a
         buffer that holds just the head and its own rows at their stream offsets; no
         rank repeats the header walk.  All buffers are allocated once.
    WHICH rows a rank decodes: NOT shard_rows(rows, world)[rank] (the encoder's split).  The ranges
    are served in the order the walk reaches them, so rank 0 -- which walks -- takes the LAST
    non-empty range and rank r >= 1 the (r - 1)-th from the top: `range_of_rank[rank]` indexes
    `parts`; `r0, r1` (block rows) and `y0, y1` (pixel rows) of THIS rank are attributes.  A caller of
    decode(..., gather=False) places its rows at y0 (decode_sharded returns them with the range).
    `trace` of the last call lists the steps in the order this rank took them (the tests
    pin that the head phase is launched before the row index is known).
    `bytes_from_rank0` of the last call = what left rank 0 (the tests bound it by
    1.2 x the stream)."""

    def __init__(self, engine, width, height, channels=4, group=None, device=None, comm_device=None,
                 stream=None, fix_t2=False):
        import torch
        import torch.distributed as dist
        self.eng, self.W, self.H, self.C, self.group, self.stream = engine, width, height, channels, group, stream
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.comm = torch.device(comm_device) if comm_device is not None else self.dev
        self.rows = (height + 7) // 8
        self.parts = shard_rows(self.rows, self.world)
        self.fix_t2 = fix_t2
        # Who decodes which row range.  The ranges are served in the order the header walk
        # reaches them, so rank 0 -- which walks, and can start on its own rows only when the walk
        # is through -- takes the LAST non-empty range and ranks 1, 2, ... the ranges from the top:
        # every remote slice is on its way while the walk is still going (the last remote one
        # at (n - 1) / n of it).  Empty ranges (more ranks than 16-row units) go to the last ranks.
        nonempty = [k for k, (a, b) in enumerate(self.parts) if b > a]
        empty = [k for k, (a, b) in enumerate(self.parts) if b <= a]
        order = ([nonempty[-1]] + nonempty[:-1] + empty) if nonempty else list(range(self.world))
        self.range_of_rank = order                      # rank -> index into self.parts
        self.serve_order = [r for r in range(1, self.world) if self.parts[order[r]][1] > self.parts[order[r]][0]]
        r0, r1 = self.parts[order[self.rank]]
        y0, y1 = min(8 * r0, height), min(8 * r1, height)
        self.r0, self.r1, self.y0, self.y1 = r0, r1, y0, y1
        dev = self.dev
        self.d_rows = torch.empty((max(y1 - y0, 1), width, channels), dtype=torch.uint8, device=dev)
        self.d_status = torch.zeros(2, dtype=torch.int32, device=dev)
        # Per-range verdicts of the header walk: allocated ONCE (the walk, on the engine and side
        # streams, writes every entry it reports; a fill on the current stream behind _fork() could
        # land on top of an early range's verdict).
        self.d_rstat = torch.zeros(16, dtype=torch.int32, device=dev)
        self.d_index = torch.zeros(2 * self.rows + 2, dtype=torch.int32, device=dev)   # offsets, lengths, first
        self.meta = torch.zeros(4 + 2 * self.rows, dtype=torch.int64, device=self.comm)
        self.d_packed = None      # grown on demand; the bytes behind the stream are zeroed per decode (_buffer)
        self.bytes_from_rank0 = 0
        self._es = None           # the engine stream (_engine_stream)

    def _engine_stream(self):
        """The HIP stream the engine's kernels of a decode run on: a stream of this decoder's
        own (or the one the caller gave), NOT torch's current stream -- the buffer fills, the
        broadcasts and the point-to-point transfers are enqueued on the current stream
        (ProcessGroupNCCL orders its own stream behind it), so a head phase launched there
        would hold back the row index and every rank's row bytes until it has finished.
        Ordering between the two is explicit: _fork() in front of a phase (the bytes it reads
        have landed), _join() behind the last one (its results are read on the current
        stream).  CPU tensors (the gloo stub tests): no streams, returns None."""
        if self.dev.type != "cuda":
            return None
        if self._es is None:
            import torch
            self._es = (torch.cuda.ExternalStream(int(self.stream), device=self.dev) if self.stream is not None
                        else torch.cuda.Stream(device=self.dev))
        return self._es

    def _s(self):
        """Raw handle of the engine stream (0 without a GPU: the stub engines ignore it)."""
        es = self._engine_stream()
        return es.cuda_stream if es is not None else 0

    def _fork(self):
        """The engine stream waits for what torch's current stream holds now."""
        es = self._engine_stream()
        if es is not None:
            import torch
            es.wait_stream(torch.cuda.current_stream(self.dev))

    def _join(self):
        """torch's current stream waits for what the engine stream holds now."""
        es = self._engine_stream()
        if es is not None:
            import torch
            torch.cuda.current_stream(self.dev).wait_stream(es)

    def _buffer(self, size):
        """The stream buffer of this rank, with [size, size rounded up to 16, + 64) zeroed:
        the row kernels read whole dwords and a few dwords ahead, rank 0 sends slices
        rounded up past the end of the stream, and neither may see a previous frame's
        bytes (a damaged last row must get the same verdict whatever was decoded before)."""
        import torch
        cap = (int(size) + 15) // 16 * 16 + 64
        if self.d_packed is None or self.d_packed.numel() < cap:
            self.d_packed = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        self.d_packed[int(size):cap].zero_()
        return self.d_packed

    def _first_rank0(self, packed):
        """-> (size, ok, rows_first, d_src or None, host array or None): the stream's size and
        where its first row header lies -- the chunk look-ups, not the header walk."""
        import torch
        import himg_amd
        if torch.is_tensor(packed) and packed.device.type != "cpu":
            size = int(packed.numel())
            src = packed
            if src.data_ptr() % 16 or src.numel() % 4:
                buf = self._buffer(size)
                buf[:size] = src
                src = buf
            self._fork()
            self.eng.decode_first_device(src, size, self.W, self.H, self.C, self.d_index[2 * self.rows:],
                                         self.d_status[1:], self._s())
            self._join()
            host = torch.cat([self.d_index[2 * self.rows: 2 * self.rows + 1], self.d_status[1:2]]).cpu().numpy()
            first = int(np.int64(host[0]) & 0xFFFFFFFF)
            return size, first != 0, first, src, None       # (a damaged container: k_dec_parse's verdict, below)
        a = packed.numpy() if torch.is_tensor(packed) else np.ascontiguousarray(packed, np.uint8)
        try:
            w, h, c, off, ln, first = himg_amd.index_host(a, self.fix_t2)
            ok = (w, h, c) == (self.W, self.H, self.C)
        except himg_amd.HimgError:
            ok, off, ln, first = False, None, None, 0
        self._host_index = (off, ln) if ok else None          # (microseconds: the whole index is there already)
        return int(a.size), ok, int(first), None, a

    def _walk_rank0(self, d_src, size):
        """Start the header walk of a stream in HBM on the engine's side stream (it runs
        beside the head phase launched after it)."""
        rows = self.rows
        self.eng.decode_walk_device(d_src, size, self.W, self.H, self.C, self.d_index, self.d_index[2 * rows:],
                                    self.d_status[1:], self._s())

    def _index_rank0(self):
        """-> (ok, offsets, lengths) once the walk is through."""
        import torch
        rows = self.rows
        self.eng.decode_walk_wait()
        host = torch.cat([self.d_index, self.d_status[1:2]]).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        return int(host[-1]) == 0, host[:rows], host[rows:2 * rows]

    def decode(self, packed=None, gather=True):
        """packed: the stream on rank 0 (numpy array, CPU or CUDA tensor), ignored
        elsewhere.  Returns (ok, pixels) like decode_sharded."""
        import torch
        import torch.distributed as dist
        world, rank, rows, group = self.world, self.rank, self.rows, self.group
        self.bytes_from_rank0 = 0
        self.trace = []
        if world == 1 and hasattr(self.eng, "decode_rows_device"):
            # One rank: nothing to scatter -- the plain row-range decode, whose header walk
            # runs beside the container parse instead of in front of it.
            size = int(packed.numel() if torch.is_tensor(packed) else len(packed))
            if torch.is_tensor(packed) and packed.device == self.dev and packed.data_ptr() % 16 == 0 \
                    and _readable_bytes(packed) >= (size + 3) // 4 * 4:
                buf = packed             # in place: the engine reads whole dwords, the storage holds them
            else:
                buf = self._buffer(size)
                src = packed if torch.is_tensor(packed) else torch.from_numpy(np.ascontiguousarray(packed, np.uint8))
                buf[:size] = src.to(self.dev)
            self.d_status.zero_()
            # (one rank: nothing travels while the engine works, so the engine stays on the caller's
            # stream -- torch's current one unless one was given -- like every torch operation around it)
            s1 = self.stream if self.stream is not None else (
                torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else 0)
            self.eng.decode_rows_device(buf, size, self.W, self.H, self.C, 0, rows, self.d_rows, self.d_status, s1)
            ok = int(self.d_status[0].item()) == 0
            if not gather:
                return ok, (self.d_rows if ok else None)
            return ok, (self.d_rows.cpu().numpy() if ok else None)
        pipelined = hasattr(self.eng, "decode_head_device")
        # ---- stage 1: the head of the stream to everybody, and what needs only the head started
        d_src = h_src = None
        m0 = torch.zeros(4, dtype=torch.int64, device=self.comm)
        if rank == 0:
            size, ok, first, d_src, h_src = self._first_rank0(packed)
            m0.copy_(torch.tensor([size, 1 if ok else 0, first, 0], dtype=torch.int64))
        if world > 1:
            dist.broadcast(m0, src=0, group=group)
            self.bytes_from_rank0 += (world - 1) * m0.numel() * 8
        size, ok, first = (int(x) for x in m0.cpu().numpy()[:3])
        self.trace.append("first")
        if not ok:
            return False, None          # the container is damaged: every rank agrees
        head16 = min((first + 15) // 16 * 16, (size + 15) // 16 * 16)
        if world == 1 and d_src is not None:
            buf = d_src                  # one rank, stream already in HBM: decode in place
        else:
            buf = self._buffer(size)
            if rank == 0 and d_src is None:
                buf[:size] = torch.from_numpy(h_src).to(self.dev)
            elif rank == 0 and d_src.data_ptr() != buf.data_ptr():
                buf[:size] = d_src[:size]
        if world > 1:
            head = buf[:head16]
            if self.comm != self.dev:
                hc = head.to(self.comm)
                dist.broadcast(hc, src=0, group=group)
                if rank != 0:
                    head.copy_(hc)
            else:
                dist.broadcast(head, src=0, group=group)
            self.bytes_from_rank0 += (world - 1) * head16
        self.trace.append("head")
        self.d_status.zero_()
        # The head of the stream is in `buf` once the current stream gets here: from this point
        # the engine's kernels run on the engine stream, BESIDE what follows on the current one
        # (the rows' index slices and bytes on their way to their owners).
        self._fork()
        ranges_walk = rank == 0 and d_src is not None and hasattr(self.eng, "decode_walk_ranges_device") \
            and len(self.serve_order) + 1 <= 16
        if ranges_walk:
            # The header walk in the row ranges of the ranks, in the order they are served: a
            # range's slice leaves as soon as the walk has passed it.
            ends = [self.parts[self.range_of_rank[r]][1] for r in self.serve_order] + [rows]
            self.eng.decode_walk_ranges_device(buf, size, self.W, self.H, self.C, ends, self.d_index,
                                               self.d_index[2 * rows:], self.d_rstat, self._s())
            self.trace.append("walk_started")
        elif rank == 0 and d_src is not None:
            self._walk_rank0(buf, size)      # (in front of the head phase: they run side by side)
            self.trace.append("walk_started")
        if pipelined:
            # Container parse, LRES chain, predictor inverse: launched now, running while rank 0
            # walks the row headers and the rows' bytes travel.
            self.eng.decode_head_device(buf, size, self.W, self.H, self.C, self._s())
            self.trace.append("head_phase")
        # ---- stage 2: every rank's index slice and row bytes, as the walk reaches them
        max_rows = max(b - a for a, b in self.parts)
        mlen = 4 + 2 * max_rows                       # [ok, lo, hi, n, offsets..., lengths...]
        my_a, my_b = self.parts[self.range_of_rank[rank]]
        idx_ok, my_off, my_len = True, None, None
        if rank == 0:
            whole = None                               # (ok, offsets, lengths) of all rows, when known at once
            if d_src is None:
                whole = (True,) + tuple(self._host_index) if self._host_index is not None else \
                    (False, np.zeros(rows, np.int64), np.zeros(rows, np.int64))
            elif not ranges_walk:
                whole = self._index_rank0()

            def index_of(k_served, a, b):
                """(ok, offsets, lengths) of rows [a, b): the k-th range the walk reaches."""
                if whole is not None:
                    return whole[0], np.asarray(whole[1][a:b], np.int64), np.asarray(whole[2][a:b], np.int64)
                self.eng.decode_walk_wait_range(k_served)
                h = torch.cat([self.d_index[a:b], self.d_index[rows + a: rows + b],
                               self.d_rstat[k_served: k_served + 1]]).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
                n = b - a
                return int(h[-1]) == 0, h[:n], h[n:2 * n]

            reqs, keep = [], []
            failed = False
            for k_served, peer in enumerate(self.serve_order):
                a, b = self.parts[self.range_of_rank[peer]]
                iok, off, ln = (False, None, None) if failed else index_of(k_served, a, b)
                failed = failed or not iok
                m = np.zeros(mlen, np.int64)
                if not failed:
                    lo, hi = slice_ranges(np.concatenate([np.zeros(a, np.int64), off]),
                                          np.concatenate([np.zeros(a, np.int64), ln]), first, size, [(a, b)])[0]
                    m[0], m[1], m[2], m[3] = 1, lo, hi, b - a
                    m[4:4 + (b - a)], m[4 + max_rows: 4 + max_rows + (b - a)] = off, ln
                mt = torch.from_numpy(m).to(self.comm)
                keep.append(mt)
                reqs.append(dist.isend(mt, peer, group=group))
                self.bytes_from_rank0 += mlen * 8
                if not failed and m[2] > m[1]:
                    lo, hi = int(m[1]), int(m[2])
                    t = buf[lo:hi] if self.comm == self.dev else buf[lo:hi].to(self.comm)
                    keep.append(t)
                    reqs.append(dist.isend(t, peer, group=group))
                    self.bytes_from_rank0 += hi - lo
                self.trace.append("slice_sent:%d" % peer)
            # ranks without rows still learn the verdict of the index
            for peer in range(1, world):
                if peer not in self.serve_order:
                    m = np.zeros(mlen, np.int64)
                    m[0] = 0 if failed else 1
                    mt = torch.from_numpy(m).to(self.comm)
                    keep.append(mt)
                    reqs.append(dist.isend(mt, peer, group=group))
                    self.bytes_from_rank0 += mlen * 8
            if my_b > my_a and not failed:
                iok, my_off, my_len = index_of(len(self.serve_order), my_a, my_b)
                failed = failed or not iok
            elif d_src is not None and ranges_walk and not failed:
                # (no rows of its own: the walk's final verdict still counts)
                self.eng.decode_walk_wait_range(len(self.serve_order))
                failed = int(self.d_rstat[len(self.serve_order)].item()) != 0
            idx_ok = not failed
            self.trace.append("index")
            for req in reqs:
                req.wait()
            self.trace.append("rows_arrived")
        else:
            mt = torch.zeros(mlen, dtype=torch.int64, device=self.comm)
            dist.recv(mt, 0, group=group)
            m = mt.cpu().numpy()
            idx_ok = bool(m[0])
            self.trace.append("index")
            if idx_ok and m[2] > m[1]:
                lo, hi, n = int(m[1]), int(m[2]), int(m[3])
                my_off, my_len = m[4:4 + n], m[4 + max_rows: 4 + max_rows + n]
                # What lies right behind this rank's slice is a previous frame's: the row kernels
                # clamp their reads to the row, but the invariant is the one-process path's
                # (himg_multi.hip zeroes 64 bytes behind every slot's slice) -- no decode ever
                # sees bytes of another frame.
                buf[hi:min(hi + 64, buf.numel())].zero_()
                if self.comm == self.dev:
                    dist.recv(buf[lo:hi], 0, group=group)
                else:
                    t = torch.empty(hi - lo, dtype=torch.uint8, device=self.comm)
                    dist.recv(t, 0, group=group)
                    buf[lo:hi] = t.to(self.dev)
            self.trace.append("rows_arrived")
        if idx_ok and my_b > my_a:
            # this rank's slice of the row index (the engine reads rows [r0, r1) of it only)
            full = np.zeros(2 * rows, np.int64)
            full[my_a:my_b], full[rows + my_a: rows + my_b] = my_off, my_len
            self.d_index[: 2 * rows] = torch.from_numpy(full.astype(np.uint32).view(np.int32)).to(self.dev)
            # The rows phase follows the head phase on the engine stream and must also see the row
            # bytes and the index, which arrived on the current stream.
            self._fork()
            if pipelined:
                self.eng.decode_rows_after_head_device(buf, size, self.W, self.H, self.C, self.r0, self.r1,
                                                       self.d_index, self.d_rows, self.d_status, self._s())
            else:
                self.eng.decode_rows_indexed_device(buf, size, self.W, self.H, self.C, self.r0, self.r1, self.d_index,
                                                    self.d_rows, self.d_status, self._s())
            self.trace.append("rows_phase")
        self._join()                         # status and pixel rows are read on the current stream
        bad = (self.d_status[:1] != 0).to(torch.int32).to(self.comm)    # (the transfer waits for the kernels)
        if not idx_ok:
            bad.fill_(1)                     # a row header is damaged: every rank agrees below
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        ok = int(bad.item()) == 0
        d_rows = self.d_rows[: self.y1 - self.y0]
        if not gather:
            return ok, (d_rows if ok else None)
        if world == 1:
            return ok, (d_rows.cpu().numpy() if ok else None)
        # Pixel rows to rank 0, exact sizes.
        H, W, C = self.H, self.W, self.C
        ops, out = [], None
        if rank == 0:
            out = torch.empty((H, W, C), dtype=torch.uint8, device=self.comm)
            if self.y1 > self.y0:
                out[self.y0:self.y1] = d_rows.to(self.comm)
            for peer in range(1, world):
                a, b = self.parts[self.range_of_rank[peer]]
                ya, yb = min(8 * a, H), min(8 * b, H)
                if yb > ya:
                    ops.append(dist.P2POp(dist.irecv, out[ya:yb], peer, group))
        elif self.y1 > self.y0:
            ops.append(dist.P2POp(dist.isend, d_rows.to(self.comm).contiguous(), 0, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if rank != 0 or not ok:
            return ok, None
        return ok, out.cpu().numpy()


def decode_sharded(engine, packed, width, height, channels=4, group=None, gather=True, device=None,
                   comm_device=None, stream=None):
    """Decode one frame with its block rows sharded over the ranks of `group`
    (ShardedDecoder; the decoder and its buffers are kept on `engine` between calls).
    Returns (ok, pixels): `ok` is the AND over the ranks (a stream the reference rejects
    is rejected); with gather=True rank 0 gets the whole H x W x C image (None
    elsewhere), otherwise every rank gets the pixel rows it decoded as a tensor -- rows
    [dec.y0, dec.y1) of the picture, where dec = engine._sharded_decoders[...] / the
    ShardedDecoder; NOT the encoder's shard_rows(rows, world)[rank] split: rank 0 holds the
    LAST non-empty range, rank r >= 1 the (r - 1)-th (see ShardedDecoder; sharded_range()
    below returns a rank's (y0, y1) without a decoder).
    `engine` provides decode_index_device / decode_rows_indexed_device
    (himg_amd.Engine); `comm_device` is where the process group can move tensors
    (CUDA for nccl/RCCL, "cpu" for gloo)."""
    fix_t2 = bool(getattr(engine, "fix_t2", False))   # the engine's HIMG_OPT_FIX_T2: rank 0's host index follows it
    key = (width, height, channels, id(group), str(device), str(comm_device), stream, fix_t2)
    cache = getattr(engine, "_sharded_decoders", None)
    if cache is None:
        cache = engine._sharded_decoders = {}
    dec = cache.get(key)
    if dec is None:
        dec = cache[key] = ShardedDecoder(engine, width, height, channels, group, device, comm_device, stream,
                                          fix_t2=fix_t2)
    return dec.decode(packed, gather)


def sharded_range(height, world, rank):
    """(y0, y1): the pixel rows rank `rank` of `world` holds after decode_sharded(..., gather=False)
    of a frame `height` pixels high -- the assignment ShardedDecoder makes (rank 0: the last
    non-empty range of shard_rows, rank r >= 1: the (r - 1)-th; empty ranges to the last ranks)."""
    rows = (height + 7) // 8
    parts = shard_rows(rows, world)
    nonempty = [k for k, (a, b) in enumerate(parts) if b > a]
    empty = [k for k, (a, b) in enumerate(parts) if b <= a]
    order = ([nonempty[-1]] + nonempty[:-1] + empty) if nonempty else list(range(world))
    r0, r1 = parts[order[rank]]
    return min(8 * r0, height), min(8 * r1, height)
