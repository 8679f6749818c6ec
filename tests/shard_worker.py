"""Worker for the multi-process sharded-encode tests (one process per rank).

    python tests/shard_worker.py <mode> <kind> <seed> <W> <H> <q> <outfile>
mode = stub : CPU only, gloo, device phases replaced by a stub built from the
              oracle's trace (checks partitioning, collectives, layout, placement)
mode = gpu  : every rank drives the HIP engine on cuda:0, collectives over gloo
              with CPU staging (the GPU box has one GPU; RCCL needs one per rank)
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import himg_amd  # noqa: E402
from himg_amd import sharded  # noqa: E402
import oracle_lib as ol  # noqa: E402


def riff_chunks(stream):
    out, idx = {}, 12
    while idx + 8 <= len(stream):
        tag = bytes(stream[idx:idx + 4]).decode("latin1")
        sz = int.from_bytes(bytes(stream[idx + 4:idx + 8]), "little")
        out[tag] = (idx + 8, sz)
        idx += 8 + sz
    return out


def row_token_hist(row):
    """Token histogram of one block row (huffman_enc.cpp:98-144), numpy."""
    h = np.zeros(261, np.int64)
    nz = np.flatnonzero(row)
    np.add.at(h, row[nz].astype(np.int64), 1)
    edges = np.concatenate([[-1], nz, [row.size]])
    runs = np.diff(edges) - 1
    for r in runs[runs > 0]:
        r = int(r)
        while r >= 16662:
            h[260] += 1
            r -= 16662
        if r == 1:
            h[0] += 1
        elif r == 2:
            h[256] += 1
        elif 3 <= r <= 6:
            h[257] += 1
        elif 7 <= r <= 22:
            h[258] += 1
        elif 23 <= r <= 278:
            h[259] += 1
        elif r >= 279:
            h[260] += 1
    return h


class StubBackend:
    """Device phases answered from the oracle's trace of the WHOLE frame."""

    def __init__(self, img, q):
        import oracle_lib as ol
        self.stream, self.tr = ol.oracle_encode(img, q, True, trace=True)
        self.rows, self.cols, self.C = self.tr["rows"], self.tr["cols"], 4
        self.row_block = self.cols * self.C * 64
        self.chunks = riff_chunks(self.stream)
        self.extra = np.array([0] * 257 + [2, 4, 8, 14], np.int64)

    def stats(self, r0, r1):
        self.r0, self.r1 = r0, r1
        sym = self.tr["fres_sym"].reshape(self.rows, self.row_block)
        self.row_hists = [row_token_hist(sym[r]) for r in range(r0, r1)]
        hist = np.sum(self.row_hists, axis=0) if r1 > r0 else np.zeros(261, np.int64)
        low = self.tr["lowres"].reshape(self.C, self.rows, self.cols)[:, r0:r1, :]
        return torch.from_numpy(hist.astype(np.int64)), torch.from_numpy(np.ascontiguousarray(low).ravel())

    def row_bits(self, hist_global):
        assert np.array_equal(hist_global.numpy(), self.tr["fres_hist"].astype(np.int64)), \
            "all-reduced histogram differs from the whole-frame histogram"
        cost = self.tr["fres_len"].astype(np.int64) + self.extra
        bits = [int((h * cost).sum()) for h in self.row_hists]
        return torch.tensor(bits, dtype=torch.int32)

    def emit(self, all_bits, start, end):
        off, sz = self.chunks["FRES"]
        rel = self.stream[off + self.tr["fres_tree_bytes"]: off + sz]
        return torch.from_numpy(rel[start:end].copy())

    def head(self, low_full, all_bits, own_start, own_end):
        """Final-placement form: the stream buffer with everything in it that rank 0 makes
        itself; what the peers send is poisoned until it arrives."""
        assert np.array_equal(low_full.numpy(), self.tr["lowres"]), "gathered low-res plane is wrong"
        off, sz = self.chunks["FRES"]
        base = off + self.tr["fres_tree_bytes"]
        buf = np.full(self.stream.size + 64, 0xAA, np.uint8)
        buf[:base] = self.stream[:base]
        buf[base + own_start: base + own_end] = self.stream[base + own_start: base + own_end]
        self._buf = torch.from_numpy(buf)
        return self._buf, base

    def finish(self, out, host=True):
        assert out.data_ptr() == self._buf.data_ptr()
        return out.numpy()[: self.stream.size].copy()

    def assemble(self, low_full, all_bits, rel_full, host=True):
        assert np.array_equal(low_full.numpy(), self.tr["lowres"]), "gathered low-res plane is wrong"
        off, sz = self.chunks["FRES"]
        head = self.stream[: off + self.tr["fres_tree_bytes"]]
        assert rel_full.numel() == sz - self.tr["fres_tree_bytes"]
        return np.concatenate([head, rel_full.numpy()])


class StubDecodeEngine:
    """decode_rows_indexed_device answered by the oracle (CPU tensors).  The rank only
    holds the head of the stream and the bytes of its own block rows; the stub checks
    exactly that against the whole stream (which every rank can regenerate), checks the
    row index it was handed, and returns block rows [r0, r1) of the oracle's picture --
    or a non-zero status when the reference rejects the stream."""

    def __init__(self, full_stream):
        self.full = np.ascontiguousarray(full_stream, np.uint8)
        self.head_done = False

    def decode_head_device(self, d_packed, size, w, h, c, stream=0):
        """The head phase: only the bytes in front of the first row header may be looked at
        (the rows' bytes have not arrived yet: the decoder poisons nothing, so check the head)."""
        full = self.full
        assert size == full.size
        first = himg_amd.index_host(full)[5] if ol.oracle_decode(full)[0] == 0 else 0
        if first:
            assert np.array_equal(d_packed.numpy()[:first], full[:first]), "head of the stream differs"
        self.head_done = True

    def decode_rows_after_head_device(self, d_packed, size, w, h, c, r0, r1, d_index, d_rows, d_status, stream=0):
        assert self.head_done, "the rows phase needs the head phase"
        self.head_done = False
        return self.decode_rows_indexed_device(d_packed, size, w, h, c, r0, r1, d_index, d_rows, d_status, stream)

    def decode_rows_indexed_device(self, d_packed, size, w, h, c, r0, r1, d_index, d_rows, d_status, stream=0):
        full, rows = self.full, (h + 7) // 8
        assert size == full.size
        _, _, _, off, ln, first = himg_amd.index_host(full)
        idx = d_index.numpy().view(np.uint32)
        # (a rank is handed the index of its OWN rows -- the slice that left rank 0 when the header
        # walk had passed them; the engine reads rows [r0, r1) of the index only)
        assert np.array_equal(idx[r0:r1], off[r0:r1]) and np.array_equal(idx[rows + r0: rows + r1], ln[r0:r1]), \
            "row index differs"
        buf = d_packed.numpy()
        assert np.array_equal(buf[:first], full[:first]), "head of the stream differs"
        if r1 > r0:
            lo, hi = int(off[r0]), int(off[r1 - 1]) + int(ln[r1 - 1])
            assert np.array_equal(buf[lo:hi], full[lo:hi]), "this rank's row bytes differ"
        rc, pix = ol.oracle_decode(full)
        if rc != 0:
            d_status[0] = 4
            return
        y0, y1 = min(8 * r0, h), min(8 * r1, h)
        if y1 > y0:
            d_rows[: y1 - y0] = torch.from_numpy(pix.reshape(h, w, c)[y0:y1].copy())


def main_decode(mode, kind, seed, W, H, q, outfile):
    """Row-sharded decode: rank 0 owns the stream; the assembled pixels (or the
    word REJECTED) go to outfile, what left rank 0 to outfile.stats."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nccl = mode == "dgpu_nccl"   # one GPU per rank, RCCL: collectives and point-to-point on device tensors
    if nccl:
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    full = ol.oracle_encode(himg_amd.synth(kind, seed, W, H), q, True)
    packed = full if rank == 0 else None
    if mode == "dstub":
        eng = StubDecodeEngine(full)
        ok, pix = sharded.decode_sharded(eng, packed, W, H, 4, device="cpu", comm_device="cpu")
    elif nccl:
        eng = himg_amd.Engine(rank)
        d_packed = torch.from_numpy(full).to(dev) if rank == 0 else None
        ok, pix = sharded.decode_sharded(eng, d_packed, W, H, 4, device=dev)
        if ok and rank == 0 and torch.is_tensor(pix):
            pix = pix.cpu().numpy()
    elif mode == "dgpu_dev":
        # The stream in rank 0's HBM (the header walk on its GPU, beside the head phase); both
        # ranks on the one GPU of the test box, collectives over gloo with CPU staging.
        eng = himg_amd.Engine(0)
        d_packed = torch.from_numpy(full).to("cuda:0") if rank == 0 else None
        ok, pix = sharded.decode_sharded(eng, d_packed, W, H, 4, device="cuda:0", comm_device="cpu")
        if rank == 0:
            assert "walk_started" in list(eng._sharded_decoders.values())[0].trace
    else:
        eng = himg_amd.Engine(0)
        ok, pix = sharded.decode_sharded(eng, packed, W, H, 4, device="cuda:0", comm_device="cpu")
    dec = list(eng._sharded_decoders.values())[0]
    if "rows_phase" in dec.trace:
        # The pipeline's order on EVERY rank: what needs only the head is launched before the
        # row index is known, the rows' decode when their bytes have arrived.
        t = dec.trace
        assert t.index("head") < t.index("head_phase") < t.index("index") < t.index("rows_arrived") < t.index("rows_phase"), t
    if rank == 0 and world > 1:
        # Rank 0 serves the row ranges in the order the header walk reaches them -- every slice
        # leaves behind the head phase's launch and in front of rank 0's own index -- and keeps
        # the LAST non-empty range for itself (its rows are the last the walk reaches).
        t = dec.trace
        sent = [e for e in t if e.startswith("slice_sent:")]
        assert sent == ["slice_sent:%d" % p for p in dec.serve_order], (sent, dec.serve_order)
        if sent:
            assert t.index("head_phase") < t.index(sent[0]) and t.index(sent[-1]) < t.index("index"), t
        nonempty = [k for k, (a, b) in enumerate(dec.parts) if b > a]
        assert dec.range_of_rank[0] == nonempty[-1]
        starts = [dec.parts[dec.range_of_rank[p]][0] for p in dec.serve_order]
        assert starts == sorted(starts) and all(s_ < dec.parts[nonempty[-1]][0] for s_ in starts)
    if rank == 0:
        if ok:
            np.asarray(pix, np.uint8).tofile(outfile)
        else:
            open(outfile, "w").write("REJECTED")
        open(str(outfile) + ".stats", "w").write("%d %d" % (dec.bytes_from_rank0, full.size))
    dist.barrier()
    dist.destroy_process_group()


def main():
    mode, kind, seed, W, H, q, outfile = sys.argv[1:8]
    seed, W, H, q = int(seed), int(W), int(H), int(q)
    if mode in ("dstub", "dgpu", "dgpu_dev", "dgpu_nccl"):
        return main_decode(mode, kind, seed, W, H, q, outfile)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nccl = mode == "gpu_nccl"   # one GPU per rank, RCCL
    if nccl:
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    img = himg_amd.synth(kind, seed, W, H)
    rows, cols = (H + 7) // 8, (W + 7) // 8
    if mode == "stub":
        backend = StubBackend(img, q)
    else:
        eng = himg_amd.Engine(rank if nccl else 0)
        r0, r1 = sharded.shard_rows(rows, world)[rank]
        # This rank only uploads its shard plus the halo (11 pixel rows above, 5 below).
        y0, y1 = max(0, 8 * r0 - 11), min(H, 8 * r1 + 5)
        if r1 <= r0:
            y0, y1 = 0, 1
        d_shard = torch.from_numpy(np.ascontiguousarray(img[y0:y1])).to("cuda:%d" % (rank if nccl else 0))
        backend = sharded.EngineBackend(eng, d_shard, y0, W, H, q, True, comm_device=None if nccl else "cpu")
    out = sharded.encode_sharded(backend, rows, cols, 4, rows > 1)
    if rank == 0:
        if torch.is_tensor(out):
            out = out.cpu().numpy()
        np.asarray(out, np.uint8).tofile(outfile)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
