#!/usr/bin/env python3
"""bench.py -- HIMG encode+decode throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: `--batch` 4096x4096 RGBA
frames (randtile, q=50; BASELINE.json configs[1]) that are ALREADY RESIDENT IN
HBM are encoded to .himg streams and decoded back, all through the C ABI
(hand-written HIP kernels).  value = pixels through encode+decode per second,
i.e. N*batch*W*H*K / wall time, whole job.  EVERY frame of every rank is a
golden input (randtile seeds 0..255): its stream and its decoded pixels are
checked against the table recorded from the real reference before anything is
timed, so a fast-but-wrong run cannot report a number.

For N > 1 there is one rank per GPU (torch.distributed, backend "nccl" = RCCL):
either the driver launches the ranks (torch.distributed.run) or, run as a plain
`python bench.py --gpus N`, this script spawns them itself BEFORE it touches the
GPU and waits for them.  Frames are independent objects, so they are sharded
over the ranks with no data-path collective (weak scaling: fixed per-GPU batch);
only the barrier and the max-over-ranks timing use the process group.  The same
line carries a `rows` object: BASELINE config 4, ONE 16384x16384 frame whose
block rows are sharded over the N ranks with the RCCL exchanges of
himg_amd/sharded.py, golden-checked (strong scaling).

Extra objects on the JSON line:
  roofline     dominant kernel, algorithmic bytes per launch / its mean duration
               measured with HIP events on the launch stream, vs 8 TB/s HBM.
  cpu_baseline the REAL reference (oracle/_ref, compiled from /root/reference)
               or, if that library is absent, our C port (oracle/), timed on this
               node's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

GOLDEN_4096 = {"packed_size": 17227700, "stream_fnv": "65c2fb5345506268",
               "decoded_fnv": "dd3685000a721519"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128,
                    help="frames per GPU per step (resident in HBM: 128 x 64 MiB of pixels, as much again of symbols "
                         "and of decoded pixels -- a tenth of the 288 GB; 64 frames: 3 percent slower, 256: 1 percent faster)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (one engine context each) the batch is split over.  Default 1: every "
                         "kernel then has the GPU to itself, so its live duration is its own (the roofline "
                         "object needs no correction); 2 streams overlap the short serial kernels of one "
                         "group with the wide kernels of the other for ~2 %% more throughput")
    ap.add_argument("--stagger", type=int, default=1,
                    help="1: odd groups run decode-then-encode so that the groups are in opposite phases")
    ap.add_argument("--width", type=int, default=None, help="default 4096 (frames) / 16384 (rows)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--quality", type=int, default=50)
    ap.add_argument("--kind", default="randtile")
    ap.add_argument("--mode", default="frames", choices=["frames", "rows"],
                    help="frames: independent frames sharded over ranks (default, the headline "
                         "metric); rows: ONE frame (default 16384x16384, BASELINE config 4) sharded by "
                         "block rows with RCCL all-reduce / all-gather / gather (encode only)")
    ap.add_argument("--rows-timeout", type=int, default=240,
                    help="seconds the config-4 rows leg may take before the frames line is printed without it")
    ap.add_argument("--no-rows", action="store_true",
                    help="skip the row-sharded 16384x16384 leg (the `rows` object of the frames line)")
    ap.add_argument("--rows-size", default=None,
                    help="TEST ONLY: WxH of the frame of the `rows` object of a frames line (default: config 4, "
                         "16384x16384 -- anything else is checked against the oracle instead of the golden table)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY (1-GPU box): every rank uses cuda:0 and the collectives run over gloo with "
                         "CPU staging, to exercise the launcher and the N > 1 orchestration; not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements (copy ceiling, latency, host API): profile runs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline time budget")
    return ap.parse_args()


def _mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def cpu_baseline(frame, quality, budget_s):
    """Time the reference CPU path on this node's host cores (rank 0, N=1 only),
    protocol of benchmark.cpp:21,111-154 (30 iterations, min / max / average) bounded
    by `budget_s`.  Test-infrastructure use of oracle/: it is the thing timed here,
    never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    h, w = frame.shape[:2]
    mpx = w * h / 1e6
    cores = os.cpu_count() or 1
    if ol.have_ref():
        kind, enc, dec = "reference", ol.ref_encode, ol.ref_decode
    else:
        kind, enc, dec = "port", ol.oracle_encode, ol.oracle_decode

    def series(fn, max_n, budget):
        ts = []
        while len(ts) < 2 or (sum(ts) < budget and len(ts) < max_n):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return ts, r
    # Encode: single thread by design (reference encoder.cpp:258-335).
    te, packed = series(lambda: enc(frame, quality, True), 30, budget_s * 0.5)
    # Decode: Decoder(0) = all hardware threads (decoder.cpp:79-85); first call warms up.
    dec(packed, 0)
    td, (rc, _) = series(lambda: dec(packed, 0), 30, budget_s * 0.25)
    assert rc == 0
    t0 = time.perf_counter()
    dec(packed, 1)
    t_dec1 = time.perf_counter() - t0
    e, d = sum(te) / len(te), sum(td) / len(td)
    # Node-level encode (SURVEY.md 8d iii): one independent single-threaded encoder per
    # core on the same frame (ctypes releases the GIL).  An encoder holds ~0.3 GB for
    # a 4096x4096 frame, so the worker count is the core count unless the node's free
    # memory says otherwise.
    from concurrent.futures import ThreadPoolExecutor
    per_worker = 5 * frame.nbytes
    mem = _mem_available_bytes()
    nw = cores if not mem else max(1, min(cores, int(0.5 * mem / per_worker)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nw) as pool:
        list(pool.map(lambda _: enc(frame, quality, True), range(nw)))
    t_node = time.perf_counter() - t0
    ms = lambda ts: {"n": len(ts), "min": round(min(ts) * 1e3, 2), "mean": round(sum(ts) / len(ts) * 1e3, 2),
                     "max": round(max(ts) * 1e3, 2)}
    return {
        "value": round(mpx / (e + d), 3), "unit": "Mpixels/s", "cores": cores, "kind": kind,
        "sample": "%dx%d RGBA %s q=%d: %d encodes on 1 thread (%.3f s each) + %d decodes on %d threads "
                  "(%.4f s each); same frame as the GPU run" % (w, h, "randtile", quality, len(te), e,
                                                                len(td), cores, d),
        "encode_ms_1thread": ms(te), "decode_ms_allthreads": ms(td),
        "encode_mpx_s_1thread": round(mpx / e, 3),
        "decode_mpx_s_allthreads": round(mpx / d, 3),
        "decode_mpx_s_1thread": round(mpx / t_dec1, 3),
        "encode_mpx_s_node": round(mpx * nw / t_node, 3), "encode_node_workers": nw,
    }


def measure_extras(torch, himg_amd, eng, dev, d_frames, d_out, d_sizes, d_st_e, d_st_d, d_pix, h_sizes,
                   frame0, W, H, Q, cap):
    """SURVEY.md 8(d) side figures (rank 0, N=1; none of them is `value`):
    an on-box copy ceiling for the HBM roofline, single-frame latency, and the
    host-buffer API whose time includes PCIe both ways."""
    out = {}
    # HBM ceiling seen by a plain device-to-device copy (read + write bytes).
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize()
    out["hbm_copy_ceiling_GBs"] = round(2.0 * n * 10 / (time.perf_counter() - t) / 1e9, 1)
    del a, b
    torch.cuda.empty_cache()
    # The same ceilings from hand-written kernels on a known byte count (himg_amd/bin/hbm_calib, built from tools/micro/hbm_calib.hip:
    # 16 bytes per lane, read only / write only / 1:1 copy, best grid of a sweep, buffers far
    # larger than L2 + Infinity Cache): what "HBM-bound" is measured against in DESIGN.md.  A
    # child process (its own HIP context; this one keeps its frames).
    exe = os.path.join(ROOT, "himg_amd", "bin", "hbm_calib")
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=120)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and line:
                c = json.loads(line[-1])
                out["hbm_ceiling_kernels_GBs"] = {"read_only": c["read_GBs"], "write_only": c["write_GBs"],
                                                  "copy_read_plus_write": c["copy_GBs_rd_plus_wr"],
                                                  "hipMemcpyDtoD_read_plus_write": c["memcpy_dtod_GBs_rd_plus_wr"],
                                                  "source": "tools/micro/hbm_calib.hip, 2 GiB buffers"}
        except (OSError, subprocess.SubprocessError, ValueError, KeyError):
            pass
    # Single-frame latency through the device-resident API (one frame per launch).
    def lat(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    out["single_frame_latency_ms"] = {
        "encode": round(lat(lambda: eng.encode_device(d_frames[:1], 1, W, H, 4, 4, Q, True, d_out[:1], cap,
                                                      d_sizes[:1], d_st_e[:1], 0)), 3),
        "decode": round(lat(lambda: eng.decode_device(d_out[:1], cap, h_sizes[:1], 1, W, H, 4, d_pix[:1],
                                                      d_st_d[:1], 0)), 3)}
    # The same decode with the block rows indexed by the caller (himg_hip_index_host on the host's
    # copy of the stream, as himg_hip_decode / Decoder::Decode do for every stream that comes from
    # host memory): the serial header walk -- 246 of the 410 us above -- does not run on the GPU.
    size0 = int(h_sizes[0])
    host0 = d_out[0, :size0].cpu().numpy()
    t = time.perf_counter()
    _, _, _, off, ln, _ = himg_amd.index_host(host0)
    t_index = time.perf_counter() - t
    d_idx = torch.from_numpy(np.concatenate([off, ln]).astype(np.uint32).view(np.int32)).to(dev)
    out["single_frame_latency_ms"]["decode_rows_indexed"] = round(lat(
        lambda: eng.decode_rows_indexed_device(d_out[0], size0, W, H, 4, 0, (H + 7) // 8, d_idx, d_pix[0],
                                               d_st_d[:1], 0)), 3)
    out["single_frame_latency_ms"]["index_host_ms"] = round(t_index * 1e3, 3)
    assert int(d_st_d[0].item()) == 0
    # Host-buffer API (himg_hip_encode / himg_hip_decode): H2D + kernels + D2H.
    # The output array is reused, like the reference benchmark reuses one Decoder
    # (benchmark.cpp:122-125): a fresh 64 MiB buffer per call would page-fault
    # for longer than the transfer takes.
    packed = eng.encode(frame0, Q, True)
    pix = eng.decode(packed)

    def reps(fn, n=20):   # per-call seconds: the mean is what is quoted, the minimum says how noisy the box is
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sum(ts) / len(ts), min(ts)
    te, te_min = reps(lambda: eng.encode(frame0, Q, True))
    td, td_min = reps(lambda: eng.decode(packed, out=pix))
    out["host_api_incl_pcie_mpx_s"] = {"encode": round(W * H / te / 1e6, 1), "decode": round(W * H / td / 1e6, 1),
                                       "encode_decode": round(W * H / (te + td) / 1e6, 1), "repetitions": 20,
                                       "best_call": {"encode": round(W * H / te_min / 1e6, 1),
                                                     "decode": round(W * H / td_min / 1e6, 1)}}
    # Batched host API: 8 frames in flight (H2D / kernels / D2H overlapped).
    nb = 8
    eouts = [np.empty(cap, np.uint8) for _ in range(nb)]
    pouts = [np.empty(W * H * 4, np.uint8) for _ in range(nb)]
    streams = eng.encode_batch([frame0] * nb, Q, True, outs=eouts)
    eng.decode_batch(streams, outs=pouts)
    t = time.perf_counter()
    streams = eng.encode_batch([frame0] * nb, Q, True, outs=eouts)
    teb = (time.perf_counter() - t) / nb
    t = time.perf_counter()
    eng.decode_batch(streams, outs=pouts)
    tdb = (time.perf_counter() - t) / nb
    out["host_batch_api_incl_pcie_mpx_s"] = {"encode": round(W * H / teb / 1e6, 1), "decode": round(W * H / tdb / 1e6, 1),
                                             "encode_decode": round(W * H / (teb + tdb) / 1e6, 1), "frames_in_flight": nb,
                                             "host_memory": "pageable"}
    # The same with the caller's buffers page-locked (himg_hip_host_alloc): true
    # asynchronous DMA both ways.
    pf = [himg_amd.pinned_empty(W * H * 4) for _ in range(nb)]
    for b_ in pf:
        b_[:] = frame0.ravel()
    pin_e = [himg_amd.pinned_empty(cap) for _ in range(nb)]
    pin_p = [himg_amd.pinned_empty(W * H * 4) for _ in range(nb)]
    fr = [b_.reshape(H, W, 4) for b_ in pf]
    streams = eng.encode_batch(fr, Q, True, outs=pin_e)
    eng.decode_batch(streams, outs=pin_p)
    t = time.perf_counter()
    streams = eng.encode_batch(fr, Q, True, outs=pin_e)
    teb = (time.perf_counter() - t) / nb
    t = time.perf_counter()
    eng.decode_batch(streams, outs=pin_p)
    tdb = (time.perf_counter() - t) / nb
    out["host_batch_api_pinned_incl_pcie_mpx_s"] = {"encode": round(W * H / teb / 1e6, 1), "decode": round(W * H / tdb / 1e6, 1),
                                                    "encode_decode": round(W * H / (teb + tdb) / 1e6, 1),
                                                    "frames_in_flight": nb, "host_memory": "pinned (himg_hip_host_alloc)"}
    return out


GOLDEN_16384 = {"packed_size": 275620945, "stream_fnv": "5bdcdb7a140df481", "decoded_fnv": "08fb9dc8e25c2fae"}


def rows_leg(args, rank, local_rank, world, dev, steps, warmup):
    """BASELINE config 4: one large frame, block rows sharded over the ranks, FRES
    histogram all-reduced, row sizes all-gathered, packed rows gathered to rank 0
    over RCCL/xGMI (himg_amd/sharded.py); then the row-sharded decode of the same
    stream.  Strong scaling: the frame is fixed.  Returns the result object on rank
    0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    import himg_amd
    from himg_amd import sharded

    if args.mode == "rows":   # on its own: the geometry can be chosen
        W, H, Q, kind = args.width or 16384, args.height or 16384, args.quality, args.kind
    else:                     # as the `rows` object of the frames line: config 4 as BASELINE.json states it
        W, H, Q, kind = 16384, 16384, 50, "randtile"
        if args.rows_size:    # (tests: the same leg on a frame that eight ranks sharing one GPU can afford)
            W, H = (int(x) for x in args.rows_size.lower().split("x"))
    rows, cols = (H + 7) // 8, (W + 7) // 8
    img = himg_amd.synth(kind, 0, W, H)               # every rank generates, then keeps its shard
    r0, r1 = sharded.shard_rows(rows, world)[rank]
    y0, y1 = max(0, 8 * r0 - 11), min(H, 8 * r1 + 5)
    if r1 <= r0:
        y0, y1 = 0, 1
    d_shard = torch.from_numpy(np.ascontiguousarray(img[y0:y1])).to(dev)
    want = None
    golden = (W, H, Q, kind) == (16384, 16384, 50, "randtile")
    if rank == 0 and not golden:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        want = ol.oracle_encode(img, Q, True)
    del img
    eng = himg_amd.Engine(local_rank)
    comm = "cpu" if args.oversubscribe else None
    backend = sharded.EngineBackend(eng, d_shard, y0, W, H, Q, True, comm_device=comm)

    def step():
        # The stream stays in rank 0's HBM (like the frames metric); the one host
        # copy below is for the bit-exactness check.
        return sharded.encode_sharded(backend, rows, cols, 4, rows > 1, host=False)

    out = step()
    verified = "n/a"
    if rank == 0:
        out = out.cpu().numpy()
        if golden:
            assert out.size == GOLDEN_16384["packed_size"], out.size
            assert himg_amd.fnv1a64(out) == GOLDEN_16384["stream_fnv"], "stream differs from the reference"
            verified = "golden"
        else:
            assert np.array_equal(out, want), "stream differs from the oracle"
            verified = "oracle"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(fn):
        """K steps inside one barrier bracket (the rate) + the per-step spread."""
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        total = time.perf_counter() - t0
        per = []
        for _ in range(steps):
            barrier()
            t = time.perf_counter()
            fn()
            barrier()
            per.append(time.perf_counter() - t)
        return total, per

    dt, per_enc = timed_steps(step)

    # Row-sharded decode of the same stream: every rank decodes its own block rows,
    # pixels stay sharded; the gathered image is hashed once against the golden
    # outside the timed region.
    d_packed = None
    if rank == 0:   # a view of a buffer padded to whole dwords: the one-rank decode then reads it in place
        pad = torch.zeros((out.size + 15) // 16 * 16 + 64, dtype=torch.uint8, device=dev)
        pad[: out.size] = torch.from_numpy(out).to(dev)
        d_packed = pad[: out.size]
    ok, pix = sharded.decode_sharded(eng, d_packed, W, H, 4, gather=True, device=dev, comm_device=comm)
    dec_verified = "n/a"
    if rank == 0:
        assert ok, "sharded decode rejected the stream"
        if golden:
            assert himg_amd.fnv1a64(pix) == GOLDEN_16384["decoded_fnv"], "pixels differ from the reference"
            dec_verified = "golden"
        else:
            rc, ref = ol.oracle_decode(want)
            assert rc == 0 and np.array_equal(pix.ravel(), ref.ravel()), "pixels differ from the oracle"
            dec_verified = "oracle"
    del pix
    dt_dec, per_dec = timed_steps(lambda: sharded.decode_sharded(eng, d_packed, W, H, 4, gather=False, device=dev,
                                                                   comm_device=comm))
    if world > 1:
        t = torch.tensor([dt, dt_dec] + per_enc + per_dec, dtype=torch.float64, device="cpu" if args.oversubscribe else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t = [float(x) for x in t.cpu()]
        dt, dt_dec, per_enc, per_dec = t[0], t[1], t[2:2 + steps], t[2 + steps:]
    eng.close()
    if rank != 0:
        return None
    spread = lambda per: {"min": round(min(per) * 1e3, 3), "mean": round(sum(per) / len(per) * 1e3, 3),
                          "max": round(max(per) * 1e3, 3)}
    enc_v, dec_v = W * H * steps / dt / 1e6, W * H * steps / dt_dec / 1e6
    return {
        "workload": "%dx%d RGBA %s q=%d: ONE frame, block rows sharded over %d rank(s); encode = RCCL all-reduce "
                    "(261-bin histogram) + all-gather (row bits) + point-to-point sends of the low-res rows and the "
                    "packed rows to rank 0 (exact sizes, received in place), stream left in rank 0's HBM; decode = "
                    "rank 0 indexes the rows once, broadcasts the head of the stream (container, LRES, FRES tree) and "
                    "sends every rank only its own rows' bytes, pixels stay sharded" % (W, H, kind, Q, world),
        "scaling": "strong", "n_gpus": world, "steps": steps,
        "encode_mpx_s": round(enc_v, 2), "decode_mpx_s": round(dec_v, 2),
        "encode_decode_mpx_s": round(W * H * steps / (dt + dt_dec) / 1e6, 2),
        "encode_ms": spread(per_enc), "decode_ms": spread(per_dec),
        "bit_exact": {"stream": verified, "pixels": dec_verified},
    }


def bench_rows(args, rank, local_rank, world, dev):
    """--mode rows: the row-sharded leg on its own, as the line's metric."""
    import torch.distributed as dist
    r = rows_leg(args, rank, local_rank, world, dev, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "metric": "Mpixels/s encode+decode, one RGBA frame row-sharded over the GPUs, q=%d" % args.quality,
            "value": r["encode_decode_mpx_s"], "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(r["encode_ms"]["mean"] + r["decode_ms"]["mean"], 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/i16/i32 integer", "data": "synthetic",
            "config": {"workload": r["workload"], "bit_exact": r["bit_exact"]}, "rows": r}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` run as a plain command: start one child per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, like
    torch.distributed.run) and wait.  This parent never touches the GPU -- it has
    not even imported torch -- so nothing that has initialised HIP is ever
    re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: be the launcher (no GPU call in this process).
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    # The runtime maps a process's streams onto 4 hardware queues by default, and streams that share
    # a queue run one after the other.  A rank of the rows leg has more than four in flight (torch's
    # stream, RCCL's, the engine stream, the context's three helper streams), and so has the one-rank
    # run once the extras have created their contexts (rows decode 2.92 ms with 4 queues, 2.80 with
    # 8; the frames line is the same either way: 85.2 / 84.9 Gpx/s).  Must be in the environment
    # before the HIP runtime starts (torch is imported below).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    import torch.distributed as dist
    import himg_amd

    if args.oversubscribe:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.oversubscribe:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.mode == "rows":
        return bench_rows(args, rank, local_rank, world, dev)

    W, H, B, Q = args.width or 4096, args.height or 4096, args.batch, args.quality
    S = max(1, min(args.streams, B))
    while B % S:
        S -= 1
    G = B // S  # frames per group = frames per kernel launch
    engines = [himg_amd.Engine(local_rank) for _ in range(S)]
    eng = engines[0]
    # Synthetic frames: rank r gets seeds r*B .. r*B+B-1 modulo 256, the extent of the
    # golden table (seed 0 is SURVEY.md's golden input).
    seeds = [(rank * B + i) % 256 for i in range(B)]
    # (Generated and uploaded frame by frame: the host never holds the batch -- eight ranks
    # of 128 frames would be 64 GiB of host memory otherwise.  Frame 0 stays for the
    # oracle check and the CPU baseline.)
    d_frames = torch.empty((B, H, W, 4), dtype=torch.uint8, device=dev)
    frames = [None]
    # The generator (a seeded xorshift chain per frame, himg_synth.c) and the golden hashes
    # below (FNV-1a, byte-serial by definition) are host work, one frame per call into the C
    # library, which drops the GIL: a pool of threads over the frames keeps the run's wall
    # time near its GPU time instead of 128 x (0.15 + 0.1) s of one core.
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max(1, min(32, (os.cpu_count() or 1) // max(1, world))))
    for i, fr in enumerate(pool.map(lambda sd: himg_amd.synth(args.kind, sd, W, H), seeds)):
        d_frames[i].copy_(torch.from_numpy(fr))
        if i == 0:
            frames[0] = fr
    cap = himg_amd.max_packed_size(W, H, 4)
    d_out = torch.empty((B, cap), dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st_e = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st_d = torch.zeros(B, dtype=torch.int32, device=dev)
    d_pix = torch.empty((B, H, W, 4), dtype=torch.uint8, device=dev)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]

    def encode():
        for i, (e, st) in enumerate(zip(engines, streams)):
            sl = slice(i * G, (i + 1) * G)
            e.encode_device(d_frames[sl], G, W, H, 4, 4, Q, True, d_out[sl], cap, d_sizes[sl],
                            d_st_e[sl], st.cuda_stream)

    # Packed sizes are needed on the host once (the decode ABI takes them as a host array).
    encode()
    torch.cuda.synchronize()
    h_sizes = d_sizes.cpu().numpy().astype(np.uint32)
    assert not d_st_e.cpu().numpy().any(), "encode failed: %s" % d_st_e.cpu().numpy()

    def decode():
        for i, (e, st) in enumerate(zip(engines, streams)):
            sl = slice(i * G, (i + 1) * G)
            e.decode_device(d_out[sl], cap, h_sizes[sl], G, W, H, 4, d_pix[sl], d_st_d[sl],
                            st.cuda_stream)

    def step():
        # Every group is encoded once and decoded once per step, each group on its own
        # stream; groups overlap.  Odd groups run decode-then-encode (they decode the
        # streams their previous step's encode left in d_out -- the same bytes, the
        # frames do not change): the groups are then in opposite phases, so the short
        # serial kernels of one (tree build, container parse, LRES chain) run beside the
        # wide kernels of the other instead of beside its own twins.
        for i, (e, st) in enumerate(zip(engines, streams)):
            sl = slice(i * G, (i + 1) * G)
            if args.stagger and (i & 1):
                e.decode_device(d_out[sl], cap, h_sizes[sl], G, W, H, 4, d_pix[sl], d_st_d[sl],
                                st.cuda_stream)
                e.encode_device(d_frames[sl], G, W, H, 4, 4, Q, True, d_out[sl], cap, d_sizes[sl],
                                d_st_e[sl], st.cuda_stream)
            else:
                e.encode_device(d_frames[sl], G, W, H, 4, 4, Q, True, d_out[sl], cap, d_sizes[sl],
                                d_st_e[sl], st.cuda_stream)
                e.decode_device(d_out[sl], cap, h_sizes[sl], G, W, H, 4, d_pix[sl], d_st_d[sl],
                                st.cuda_stream)

    decode()
    torch.cuda.synchronize()
    assert not d_st_d.cpu().numpy().any(), "decode failed: %s" % d_st_d.cpu().numpy()

    # Parity gate before any timing counts: EVERY frame of this rank's batch, stream
    # and decoded pixels, against the table recorded from the real reference
    # (tests/golden/batch_4096x4096_q50.json, seeds 0..255).
    verified = "n/a"
    table = None
    tpath = os.path.join(ROOT, "tests", "golden", "batch_%dx%d_q%d.json" % (W, H, Q))
    if args.kind == "randtile" and os.path.exists(tpath):
        table = json.load(open(tpath))["seeds"]
    if table is not None and len(table) >= 256:
        def check(i):
            want_size, want_s, want_p = table[seeds[i]]
            assert int(h_sizes[i]) == want_size, (rank, i, int(h_sizes[i]), want_size)
            assert himg_amd.fnv1a64(d_out[i, :want_size].cpu().numpy()) == want_s, \
                "stream of frame %d (rank %d) differs from the reference" % (i, rank)
            assert himg_amd.fnv1a64(d_pix[i].cpu().numpy()) == want_p, \
                "pixels of frame %d (rank %d) differ from the reference" % (i, rank)
        list(pool.map(check, range(B)))   # (re-raises the first failure)
        verified = "golden, all %d frames of every rank (stream + pixels)" % B
    elif rank == 0:
        # Other geometries / qualities: frame 0 against the oracle (CPU seconds per frame).
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        want = ol.oracle_encode(frames[0], Q, True)
        assert np.array_equal(d_out[0, : int(h_sizes[0])].cpu().numpy(), want), "stream differs from the oracle"
        rc, ref = ol.oracle_decode(want)
        assert rc == 0 and np.array_equal(d_pix[0].cpu().numpy().ravel(), ref.ravel()), "pixels differ from the oracle"
        verified = "oracle, frame 0 (stream + pixels)"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    for e in engines:
        e.profile_reset()
        e.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = {}
    for e in engines:
        e.profile(False)
        for k, v in e.profile_read().items():  # stage -> (total ms, launches), HIP events on its stream
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + v[0], a[1] + v[1])

    # Per-step spread (benchmark.cpp:151-154 reports Min / Max / Average over 30
    # iterations): the same step, synchronised after each one, >= 30 times.
    per_step = []
    for _ in range(max(30, args.steps)):
        barrier()
        t = time.perf_counter()
        step()
        barrier()
        per_step.append(time.perf_counter() - t)

    # Encode-only and decode-only rates (same protocol, not part of `value`).
    def timed(fn, n):
        barrier()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n
    n_side = max(3, args.steps // 2)
    t_enc = timed(encode, n_side)
    t_dec = timed(decode, n_side)

    # The same kernels with nothing else on the GPU (group 0 on its stream only):
    # under the multi-stream schedule above a kernel's event duration includes the
    # time it shares the CUs with the other groups' kernels.
    iso = {}
    if rank == 0:
        e0, st0, sl0 = engines[0], streams[0], slice(0, G)
        barrier_local = torch.cuda.synchronize
        barrier_local()
        e0.profile_reset()
        e0.profile(True)
        for _ in range(3):
            e0.encode_device(d_frames[sl0], G, W, H, 4, 4, Q, True, d_out[sl0], cap, d_sizes[sl0],
                             d_st_e[sl0], st0.cuda_stream)
            e0.decode_device(d_out[sl0], cap, h_sizes[sl0], G, W, H, 4, d_pix[sl0], d_st_d[sl0],
                             st0.cuda_stream)
        barrier_local()
        e0.profile(False)
        iso = {k: v[0] / max(v[1], 1) for k, v in e0.profile_read().items()}

    if world > 1:
        t = torch.tensor([dt, t_enc, t_dec] + per_step, dtype=torch.float64, device="cpu" if args.oversubscribe else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t = [float(x) for x in t.cpu()]
        dt, t_enc, t_dec, per_step = t[0], t[1], t[2], t[3:]

    out = None
    if rank == 0:
        px_step = float(world) * B * W * H
        ms_step = dt / args.steps * 1e3
        value = px_step * args.steps / dt / 1e6
        packed_total = float(h_sizes.astype(np.float64).sum())
        # algorithmic bytes per pixel-frame = W*H*4 read + packed written
        # (encode), packed read + W*H*4 written (decode); one launch processes B frames.
        alg_bytes_side = B * W * H * 4.0 + packed_total
        alg_bytes_launch = alg_bytes_side * G / B   # one launch processes G = B/streams frames
        enc_stages = {"k_lowres_avg", "k_lowres_blend", "k_lres_predict", "k_tile_fwd", "k_pix_fwd", "k_lres_summary",
                      "k_tok_hist", "k_tok", "k_tree", "k_sizes", "k_emit", "k_emit_tok", "k_padfix", "memset"}
        # Per encode / decode call of one group (a kernel launched twice per call --
        # k_tok_hist: LRES spans, then FRES rows -- counts with both launches).
        calls = max(1, args.steps * len(engines))
        stages = {k: {"ms": v[0] / calls, "launches": v[1] / calls} for k, v in prof.items()}
        dom = max(stages, key=lambda k: stages[k]["ms"]) if stages else None
        roofline = None
        if dom:
            ach = alg_bytes_launch / (stages[dom]["ms"] * 1e-3) / 1e9
            # HBM bytes per launch: NOT measured in this run -- rocprofv3 cannot wrap its
            # own host.  They come from the committed PMC passes of the same workload
            # (tools/profile_pmc.sh -> profiles/traffic.json, which names the commit it was
            # measured at); null when the workload differs.
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            std = (W, H, Q, args.kind) == (4096, 4096, 50, "randtile")
            key = dom.strip("()").split("<")[0]
            pmc = {}
            dyn_src = None
            if os.path.exists(tpath) and std:
                tj = json.load(open(tpath))
                per_frame = {k.split("<")[0]: v for k, v in tj.get("bytes_per_frame", {}).items()}
                per_launch = {k.split("<")[0]: v for k, v in tj.get("bytes_per_launch", {}).items()}
                if key in per_launch and tj.get("frames_per_launch") == G:
                    traffic = per_launch[key]          # read: the PMC passes ran this launch size
                    traffic_src = "profiles/traffic.json (PMC passes at commit %s with %d frames per launch, not this run)" % (
                        tj.get("git_sha", "?"), G)
                elif key in per_frame:
                    traffic = per_frame[key] * G       # scaled from %d frames per launch
                    traffic_src = "profiles/traffic.json (PMC passes at commit %s with %s frames per launch, scaled; not this run)" % (
                        tj.get("git_sha", "?"), tj.get("frames_per_launch", "?"))
                pmc = tj.get("valu", {})
                # The three kernels whose hot loops carry trip counters (tools/dynamic_mix.py): the
                # class-weighted cost taken over what they EXECUTE instead of over their static mix.
                # (the newest round's file: the mix must describe the kernels that were timed)
                import glob
                dfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_dynamic_mix.json")))
                dpath = dfiles[-1] if dfiles else ""
                dyn_src = os.path.basename(dpath) if dpath else None
                if dpath:
                    for k, v in json.load(open(dpath)).get("kernels", {}).items():
                        if k in pmc:
                            pmc[k]["mean_cost_per_valu_dynamic"] = v["mean_cost_per_valu_dynamic"]
                            pmc[k]["issue_frac_dynamic_mix"] = v["issue_frac_dynamic_mix"]
                            pmc[k]["executed_valu_shares"] = {p_["part"].split(" (")[0].split(":")[0]: p_["share_of_executed_valu"]
                                                             for p_ in v["parts"]}
            # VALU issue of that kernel from the same PMC passes: the SIMD issue time of its
            # instruction count at the kernel's class-weighted measured cost
            # (profiles/r04_isa_mix.json over profiles/r03_valu_rate.txt) over its duration.
            valu_busy = None
            for k, v in pmc.items():
                if k.split("<")[0] == key:
                    valu_busy = v.get("issue_frac_dynamic_mix", v.get("issue_frac_measured_mix", v.get("issue_frac")))
            roofline = {"bound": "hbm", "kernel": dom, "side": "encode" if dom.strip("()").split("<")[0] in enc_stages else "decode",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": alg_bytes_launch, "frames_per_launch": G,
                        "kernel_ms": round(stages[dom]["ms"], 4), "valu_issue_frac_pmc": valu_busy}
            if dom in iso and iso[dom] > 0:
                a_iso = alg_bytes_launch / (iso[dom] * 1e-3) / 1e9
                roofline["isolated"] = {"kernel_ms": round(iso[dom], 4), "achieved": round(a_iso, 1),
                                        "frac": round(a_iso / HBM_PEAK_GBS, 4),
                                        "note": "same kernel, same launch size, no other stream running"}
        out = {
            "metric": "Mpixels/s encode+decode, 4K RGBA q=50", "value": round(value, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "step_ms": {"n": len(per_step), "min": round(min(per_step) * 1e3, 3),
                        "mean": round(sum(per_step) / len(per_step) * 1e3, 3), "max": round(max(per_step) * 1e3, 3),
                        "note": "each step synchronised on its own (not the timed region of `value`)"},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i16/i32 integer", "data": "synthetic",
            "config": {"workload": "%dx%d RGBA %s q=%d, encode+decode, batch %d frames/GPU resident in HBM, "
                                   "%d stream(s) x %d frames per launch" % (W, H, args.kind, Q, B, S, G),
                       "parallelism": "independent frames sharded over %d rank(s), no data-path collective" % world,
                       "bit_exact": verified},
            "encode_mpx_s": round(world * B * W * H / t_enc / 1e6, 2),
            "decode_mpx_s": round(world * B * W * H / t_dec / 1e6, 2),
            "encode_read_roofline_frac": round(B * W * H * 4.0 / t_enc / 1e9 / HBM_PEAK_GBS, 4),
            "pipeline_roofline": {
                "encode_GBs": round(alg_bytes_side / t_enc / 1e9, 1),
                "decode_GBs": round(alg_bytes_side / t_dec / 1e9, 1),
                "encode_frac": round(alg_bytes_side / t_enc / 1e9 / HBM_PEAK_GBS, 4),
                "decode_frac": round(alg_bytes_side / t_dec / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": roofline,
            "roofline_valu": ({"note": "per kernel: SQ_INSTS_VALU per symbol and the fraction of its duration the "
                                       "1024 SIMDs need to ISSUE that count -- at the guide's 2 cycles per wave64 "
                                       "instruction (SIMD-32) and at the kernel's class-weighted measured cost "
                                       "(plain VOP1/VOP2 ~2.2, VOP3 / packed / DPP / compares ~4.1 cycles: "
                                       "profiles/r03_valu_rate.txt, the newest profiles/rNN_isa_mix.json); "
                                       "issue_frac_dynamic_mix where the kernel's hot loops carry trip counters: "
                                       "the cost weighted by what is EXECUTED (measured trip counts x each loop's "
                                       "hot path, `dynamic_mix_source`) instead of by the static mix",
                               "source": traffic_src, "dynamic_mix_source": dyn_src, "kernels": pmc} if pmc else None),
            "stages_ms": {k: round(v["ms"], 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if world == 1 and not args.no_extras:
            out["extras"] = measure_extras(torch, himg_amd, eng, dev, d_frames, d_out, d_sizes, d_st_e,
                                           d_st_d, d_pix, h_sizes, frames[0], W, H, Q, cap)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames[0], Q, args.cpu_seconds)
    for e in engines:
        e.close()
    # BASELINE config 4 on the same ranks: one 16384x16384 frame, block rows sharded
    # over them, the RCCL exchanges of himg_amd/sharded.py (every rank takes part).
    if not args.no_rows:
        del d_frames, d_out, d_pix, frames
        torch.cuda.empty_cache()
        # The frames line must come out whatever happens to the rows leg (its RCCL
        # exchanges are the only collectives in this program): if the leg raises, or
        # has not finished within --rows-timeout seconds (a rank stuck in a collective
        # its peer left), rank 0 prints the line with the reason in "rows" and every
        # rank leaves.
        import threading

        # Neither way out is a success: the line is printed so that the frames result is not
        # lost, and the process ends NON-ZERO -- a rank hung in a collective or a stream
        # that differs from the reference must not look like a clean run.
        def give_up():
            if rank == 0:
                out["rows"] = {"error": "rows leg did not finish within %d s" % args.rows_timeout}
                print(json.dumps(out), flush=True)
            os._exit(4 if rank == 0 else 3)

        dog = threading.Timer(args.rows_timeout + (0 if rank == 0 else 5), give_up)
        dog.daemon = True
        dog.start()
        try:
            # (30 untimed calls of each side first: the leg follows the extras and the CPU baseline, half
            # a minute in which the GPU idled; with two calls the first timed steps ran at idle clocks --
            # decode 2.94 ms where the same binary gives 2.67 straight after the frames leg)
            rows_obj = rows_leg(args, rank, local_rank, world, dev, max(5, min(args.steps, 10)), 30)
        except Exception as e:   # noqa: BLE001 -- reported in the line, not swallowed
            if rank != 0:
                raise
            rows_obj = {"error": "%s: %s" % (type(e).__name__, e)}
            out["rows"] = rows_obj
            print(json.dumps(out), flush=True)
            os._exit(4)
        dog.cancel()
        if rank == 0:
            out["rows"] = rows_obj
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
