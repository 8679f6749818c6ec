"""himg_amd -- MI355X-native HIMG encode/decode engine (Python plumbing).

The product is the C-ABI shared library (include/himg_hip.h) built from the
hand-written HIP kernels under himg_amd/csrc/.  This module is only the ctypes
binding that tests, bench.py and the multi-GPU driver use; PyTorch supplies
device memory, streams and torch.distributed, nothing else.

There is no CPU fallback: if the native library cannot be loaded or no GPU is
usable, the compute entry points raise.
"""
import ctypes as C
import os

import numpy as np

from .build import LIB, build_lib

HIMG_OK = 0
HIMG_ERR_ARG = -1
HIMG_ERR_HIP = -2
HIMG_ERR_UNSUPPORTED = -3
HIMG_ERR_FORMAT = -4
HIMG_ERR_CAPACITY = -5

SYNTH = {"grad": 0, "gradn": 1, "rand": 2, "randtile": 3}

# himg_hip_debug_read selectors (include/himg_hip.h)
DBG = {
    "avg": 0, "lowres": 1, "lres_sym": 2, "fres_sym": 3, "lres_hist": 4, "fres_hist": 5,
    "lres_len": 6, "fres_len": 7, "lres_code": 8, "fres_code": 9, "fres_row_bytes": 10,
    "dec_stats": 11, "parse_stats": 12, "rowcount_stats": 13, "loop_counts": 14, "fres_tok_sym": 15,
}
DBG_DECODER = 0x100

_lib = None


class HimgError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("himg_hip error %d %s" % (code, msg))
        self.code = code


def lib():
    """Load (building if necessary) the native library; fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB
    if not os.path.exists(path):
        path = build_lib()
    # PyTorch bundles its own HIP runtime (same SONAME as /opt/rocm's).  Whichever
    # copy is loaded first serves the whole process, and torch cannot enumerate
    # GPUs through a foreign copy -- so when torch is installed let it load first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, i32, u32, u64, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_size_t
    P = C.POINTER
    L.himg_hip_create.argtypes = [i32, P(vp)]
    L.himg_hip_destroy.argtypes = [vp]
    L.himg_hip_destroy.restype = None
    L.himg_hip_last_error.argtypes = [vp]
    L.himg_hip_last_error.restype = C.c_char_p
    L.himg_hip_max_packed_size.argtypes = [i32, i32, i32]
    L.himg_hip_max_packed_size.restype = sz
    L.himg_hip_encode.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, P(vp), P(sz)]
    L.himg_hip_decode.argtypes = [vp, vp, sz, P(vp), P(i32), P(i32), P(i32)]
    L.himg_hip_encode_to.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, P(sz)]
    L.himg_hip_decode_to.argtypes = [vp, vp, sz, vp, sz, P(i32), P(i32), P(i32)]
    L.himg_hip_fetch_last.argtypes = [vp, vp, sz, P(sz)]
    L.himg_hip_peek.argtypes = [vp, sz, P(i32), P(i32), P(i32)]
    L.himg_hip_set_option.argtypes = [vp, i32, i32]
    L.himg_hip_get_option.argtypes = [vp, i32, C.POINTER(C.c_int)]
    L.himg_hip_encode_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.himg_hip_decode_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.himg_hip_free.argtypes = [vp]
    L.himg_hip_free.restype = None
    L.himg_hip_host_alloc.argtypes = [sz]
    L.himg_hip_host_alloc.restype = vp
    L.himg_hip_host_free.argtypes = [vp]
    L.himg_hip_host_free.restype = None
    L.himg_hip_encode_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp, vp, vp]
    L.himg_hip_decode_device.argtypes = [vp, vp, sz, vp, i32, i32, i32, i32, vp, vp, vp]
    L.himg_hip_decode_rows_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.himg_hip_decode_index_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, vp, vp, vp, vp]
    L.himg_hip_decode_rows_indexed_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.himg_hip_decode_rows_after_head_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.himg_hip_decode_head_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, vp]
    L.himg_hip_decode_first_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, vp, vp, vp]
    L.himg_hip_decode_walk_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, vp, vp, vp, vp]
    L.himg_hip_decode_walk_wait.argtypes = [vp]
    L.himg_hip_decode_walk_ranges_device.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, C.POINTER(C.c_int), i32, vp, vp, vp, vp]
    L.himg_hip_decode_walk_wait_range.argtypes = [vp, i32]
    L.himg_hip_index_host.argtypes = [vp, sz, i32, P(i32), P(i32), P(i32), vp, sz, P(C.c_uint32)]
    L.himg_hip_create_multi.argtypes = [vp, i32, P(vp)]
    L.himg_hip_destroy_multi.argtypes = [vp]
    L.himg_hip_destroy_multi.restype = None
    L.himg_hip_multi_count.argtypes = [vp]
    L.himg_hip_multi_last_error.argtypes = [vp]
    L.himg_hip_multi_last_error.restype = C.c_char_p
    L.himg_hip_multi_set_option.argtypes = [vp, i32, i32]
    L.himg_hip_multi_encode_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.himg_hip_multi_decode_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.himg_hip_multi_encode.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, P(vp), P(sz)]
    L.himg_hip_multi_decode.argtypes = [vp, vp, sz, P(vp), P(i32), P(i32), P(i32)]
    L.himg_hip_shard_stats.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.himg_hip_shard_row_bits.argtypes = [vp, vp, vp, vp]
    L.himg_hip_shard_emit.argtypes = [vp, vp, vp, sz, vp, vp]
    L.himg_hip_shard_assemble.argtypes = [vp, vp, vp, vp, sz, vp, sz, vp, vp, vp]
    L.himg_hip_shard_head.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, vp]
    L.himg_hip_shard_finish.argtypes = [vp, vp, sz, vp, vp]
    L.himg_hip_debug_read.argtypes = [vp, i32, i32, vp, sz, P(sz)]
    L.himg_hip_profile_enable.argtypes = [vp, i32]
    L.himg_hip_profile_reset.argtypes = [vp]
    L.himg_hip_profile_read.argtypes = [vp, P(i32), P(C.c_char_p), P(C.c_double), P(i32)]
    L.himg_synth_fill.argtypes = [i32, u64, i32, i32, vp]
    L.himg_fnv1a64.argtypes = [vp, sz]
    L.himg_fnv1a64.restype = u64
    L.himg_tables_shift.argtypes = [i32, i32, vp]
    L.himg_tables_shift.restype = None
    L.himg_tables_lowres_map.argtypes = [i32, vp]
    L.himg_tables_lowres_map.restype = None
    L.himg_tables_fullres_map.argtypes = [vp]
    L.himg_tables_fullres_map.restype = None
    L.himg_tables_map_to_8bit.argtypes = [vp, i32]
    L.himg_tables_map_to_8bit.restype = C.c_uint8
    _lib = L
    return L


# ---- host utilities (no GPU) -------------------------------------------------

def synth(kind, seed, width, height):
    """Synthetic RGBA frame as a (H, W, 4) uint8 array."""
    a = np.empty((height, width, 4), np.uint8)
    rc = lib().himg_synth_fill(SYNTH[kind], seed, width, height, a.ctypes.data)
    if rc:
        raise HimgError(rc, "synth")
    return a


def fnv1a64(buf):
    b = np.ascontiguousarray(np.frombuffer(buf, np.uint8) if isinstance(buf, (bytes, bytearray)) else buf)
    return "%016x" % lib().himg_fnv1a64(b.ctypes.data, b.nbytes)


def max_packed_size(width, height, channels):
    return int(lib().himg_hip_max_packed_size(width, height, channels))


def pinned_empty(nbytes):
    """uint8 numpy array in page-locked host memory (himg_hip_host_alloc): the host
    API's transfers from / to it are asynchronous DMA.  The allocation is released when
    the ctypes buffer behind the array -- which every view of it keeps alive -- is
    collected."""
    import weakref
    ptr = lib().himg_hip_host_alloc(int(nbytes))
    if not ptr:
        raise MemoryError("himg_hip_host_alloc(%d)" % nbytes)
    buf = (C.c_uint8 * int(nbytes)).from_address(ptr)
    weakref.finalize(buf, lib().himg_hip_host_free, ptr)
    return np.frombuffer(buf, np.uint8)


def psnr(a, b):
    """PSNR over all channels, 10*log10(255^2/MSE), double precision."""
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)


# ---- engine ------------------------------------------------------------------

class Engine:
    """One C-ABI context (one device).  Mirrors the reference's Encoder/Decoder
    pair: encode()/decode() take and return host buffers like
    himg::Encoder::Encode / himg::Decoder::Decode; the *_device methods work on
    HBM-resident batches (torch CUDA tensors or raw device pointers)."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        rc = lib().himg_hip_create(device, C.byref(self._ctx))
        if rc:
            raise HimgError(rc, "himg_hip_create (no usable GPU? there is no CPU fallback)")
        self.device = device
        # What the context took from the environment (HIMG_FIX_T2=1) counts too: the row-sharded
        # decoder's host index must follow the engine's own rule (sharded.py).
        self.fix_t2 = bool(self.get_option("fix_t2"))

    def close(self):
        if self._ctx:
            lib().himg_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise HimgError(rc, "%s: %s" % (what, lib().himg_hip_last_error(self._ctx).decode()))

    # host-buffer API ---------------------------------------------------------
    def encode(self, img, quality=50, use_ycbcr=True, channels=None, pixel_stride=None):
        """himg_hip_encode_to + himg_hip_fetch_last: the stream is fetched into an
        array of exactly its size (no intermediate copies)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        ch = channels if channels is not None else (img.shape[2] if img.ndim == 3 else 1)
        stride = pixel_stride if pixel_stride is not None else (img.shape[2] if img.ndim == 3 else 1)
        n = C.c_size_t()
        rc = lib().himg_hip_encode_to(self._ctx, img.ctypes.data, w, h, stride, ch, quality,
                                      1 if use_ycbcr else 0, None, 0, C.byref(n))
        if rc != HIMG_ERR_CAPACITY or n.value == 0:
            self._check(rc if rc != HIMG_OK else HIMG_ERR_ARG, "encode")
        out = np.empty(n.value, np.uint8)
        self._check(lib().himg_hip_fetch_last(self._ctx, out.ctypes.data, out.nbytes, C.byref(n)), "encode")
        return out

    def decode(self, packed, out=None):
        """himg_hip_peek + himg_hip_decode_to straight into `out` (reused when it has
        the right size) or a new array."""
        packed = np.ascontiguousarray(np.frombuffer(packed, np.uint8) if isinstance(packed, (bytes, bytearray)) else packed)
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        dst, cap = None, 0
        if lib().himg_hip_peek(packed.ctypes.data, packed.nbytes, C.byref(w), C.byref(h), C.byref(c)) == HIMG_OK:
            n = w.value * h.value * c.value
            if out is None or out.nbytes != n or not out.flags["C_CONTIGUOUS"] or out.dtype != np.uint8:
                out = np.empty(n, np.uint8)
            dst, cap = out.ctypes.data, out.nbytes
        rc = lib().himg_hip_decode_to(self._ctx, packed.ctypes.data, packed.nbytes, dst, cap,
                                      C.byref(w), C.byref(h), C.byref(c))
        self._check(rc, "decode")
        return out.reshape(h.value, w.value, c.value)

    def encode_batch(self, frames, quality=50, use_ycbcr=True, outs=None):
        """himg_hip_encode_batch: frames of one geometry, transfers overlapped with the
        kernels.  `outs` (optional) are reusable uint8 buffers of at least
        max_packed_size bytes; returns the streams (views into outs when given)."""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        n = len(frames)
        h, w = frames[0].shape[:2]
        ch = frames[0].shape[2] if frames[0].ndim == 3 else 1
        cap = max_packed_size(w, h, ch)
        if outs is None:
            outs = [np.empty(cap, np.uint8) for _ in range(n)]
        src = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
        sizes = (C.c_size_t * n)()
        rc = lib().himg_hip_encode_batch(self._ctx, src, n, w, h, ch, ch, quality, 1 if use_ycbcr else 0,
                                         dst, caps, sizes)
        self._check(rc, "encode_batch")
        return [o[: sizes[i]] for i, o in enumerate(outs)]

    def decode_batch(self, streams, outs=None):
        """himg_hip_decode_batch: returns the decoded frames; `outs` (optional) are
        reusable uint8 buffers large enough for the pixels."""
        streams = [np.ascontiguousarray(np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else s)
                   for s in streams]
        n = len(streams)
        if outs is None:
            outs = []
            for s_ in streams:
                w, h, c = C.c_int(), C.c_int(), C.c_int()
                ok = lib().himg_hip_peek(s_.ctypes.data, s_.nbytes, C.byref(w), C.byref(h), C.byref(c)) == HIMG_OK
                outs.append(np.empty(w.value * h.value * c.value if ok else 1, np.uint8))
        src = (C.c_void_p * n)(*[s_.ctypes.data for s_ in streams])
        szs = (C.c_size_t * n)(*[s_.nbytes for s_ in streams])
        dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
        ws, hs, cs = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
        rc = lib().himg_hip_decode_batch(self._ctx, src, szs, n, dst, caps, ws, hs, cs)
        self._check(rc, "decode_batch")
        return [o.ravel()[: ws[i] * hs[i] * cs[i]].reshape(hs[i], ws[i], cs[i]) for i, o in enumerate(outs)]

    def get_option(self, option):
        """himg_hip_get_option: the option as the context holds it (names as in set_option)."""
        opt = {"fix_t2": 1, "count_wave": 2, "emit_rows": 3, "row_tokens": 4, "front": 5}[option] if isinstance(option, str) else int(option)
        v = C.c_int(0)
        self._check(lib().himg_hip_get_option(self._ctx, opt, C.byref(v)), "get_option")
        return v.value

    def set_option(self, option, value):
        """himg_hip_set_option; option names: "fix_t2", and the kernel-variant selectors
        "count_wave" / "emit_rows" (-1 = by launch size, 0 / 1 = force; see include/himg_hip.h)."""
        opt = {"fix_t2": 1, "count_wave": 2, "emit_rows": 3, "row_tokens": 4, "front": 5}[option] if isinstance(option, str) else int(option)
        self._check(lib().himg_hip_set_option(self._ctx, opt, int(value)), "set_option")
        if opt == 1:
            self.fix_t2 = bool(value)   # (the row-sharded decoder's host index follows it, sharded.py)

    # device-resident API -------------------------------------------------------
    def encode_device(self, d_frames, batch, width, height, pixel_stride, channels, quality,
                      use_ycbcr, d_out, out_stride, d_sizes, d_status, stream=0):
        rc = lib().himg_hip_encode_device(self._ctx, _ptr(d_frames), batch, width, height,
                                          pixel_stride, channels, quality, 1 if use_ycbcr else 0,
                                          _ptr(d_out), out_stride, _ptr(d_sizes), _ptr(d_status),
                                          C.c_void_p(stream))
        self._check(rc, "encode_device")

    def decode_device(self, d_packed, in_stride, h_sizes, batch, width, height, channels, d_out,
                      d_status, stream=0):
        hs = np.ascontiguousarray(h_sizes, np.uint32)
        rc = lib().himg_hip_decode_device(self._ctx, _ptr(d_packed), in_stride, hs.ctypes.data,
                                          batch, width, height, channels, _ptr(d_out),
                                          _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "decode_device")

    def decode_rows_device(self, d_packed, packed_size, width, height, channels, row0, row1,
                           d_out_rows, d_status, stream=0):
        """Block rows [row0, row1) of one frame (row-sharded decode, himg_amd/sharded.py)."""
        rc = lib().himg_hip_decode_rows_device(self._ctx, _ptr(d_packed), int(packed_size), width,
                                               height, channels, row0, row1, _ptr(d_out_rows),
                                               _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "decode_rows_device")

    def decode_index_device(self, d_packed, packed_size, width, height, channels, d_row_index,
                            d_rows_first, d_status, stream=0):
        """Row index of a stream in HBM: [rows] payload offsets + [rows] lengths (uint32) and
        the offset of the first row header (rank 0 of a row-sharded decode)."""
        rc = lib().himg_hip_decode_index_device(self._ctx, _ptr(d_packed), int(packed_size), width, height,
                                                channels, _ptr(d_row_index), _ptr(d_rows_first),
                                                _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "decode_index_device")

    def decode_rows_indexed_device(self, d_packed, packed_size, width, height, channels, row0, row1,
                                   d_row_index, d_out_rows, d_status, stream=0):
        """Block rows [row0, row1) from a buffer that holds only the bytes in front of the first
        row header and these rows' payloads, with the row index supplied."""
        rc = lib().himg_hip_decode_rows_indexed_device(self._ctx, _ptr(d_packed), int(packed_size), width,
                                                       height, channels, row0, row1, _ptr(d_row_index),
                                                       _ptr(d_out_rows), _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "decode_rows_indexed_device")

    def decode_head_device(self, d_packed, packed_size, width, height, channels, stream=0):
        """What needs only the head of the stream (container parse, LRES chain, predictor
        inverse); decode_rows_after_head_device follows on the same stream."""
        rc = lib().himg_hip_decode_head_device(self._ctx, _ptr(d_packed), int(packed_size), width, height,
                                               channels, C.c_void_p(stream))
        self._check(rc, "decode_head_device")

    def decode_rows_after_head_device(self, d_packed, packed_size, width, height, channels, row0, row1,
                                      d_row_index, d_out_rows, d_status, stream=0):
        rc = lib().himg_hip_decode_rows_after_head_device(self._ctx, _ptr(d_packed), int(packed_size), width,
                                                          height, channels, row0, row1, _ptr(d_row_index),
                                                          _ptr(d_out_rows), _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "decode_rows_after_head_device")

    def decode_first_device(self, d_packed, packed_size, width, height, channels, d_rows_first, d_status,
                            stream=0):
        """Offset of the first FRES row header of a stream in HBM, without the header walk."""
        rc = lib().himg_hip_decode_first_device(self._ctx, _ptr(d_packed), int(packed_size), width, height,
                                                channels, _ptr(d_rows_first), _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "decode_first_device")

    def decode_walk_device(self, d_packed, packed_size, width, height, channels, d_row_index, d_rows_first,
                           d_status, stream=0):
        """The row index by the header walk alone, on the context's side stream (beside a head
        phase launched after this call); results valid after decode_walk_wait()."""
        rc = lib().himg_hip_decode_walk_device(self._ctx, _ptr(d_packed), int(packed_size), width, height, channels,
                                               _ptr(d_row_index), _ptr(d_rows_first), _ptr(d_status),
                                               C.c_void_p(stream))
        self._check(rc, "decode_walk_device")

    def decode_walk_ranges_device(self, d_packed, packed_size, width, height, channels, range_end, d_row_index,
                                  d_rows_first, d_range_status, stream=0):
        """himg_hip_decode_walk_ranges_device: the header walk in row ranges on the side stream;
        range k's index entries and verdict are complete after decode_walk_wait_range(k)."""
        ends = (C.c_int * len(range_end))(*[int(x) for x in range_end])
        rc = lib().himg_hip_decode_walk_ranges_device(self._ctx, _ptr(d_packed), int(packed_size), width, height,
                                                      channels, ends, len(range_end), _ptr(d_row_index),
                                                      _ptr(d_rows_first), _ptr(d_range_status), C.c_void_p(stream))
        self._check(rc, "decode_walk_ranges_device")

    def decode_walk_wait_range(self, k):
        self._check(lib().himg_hip_decode_walk_wait_range(self._ctx, int(k)), "decode_walk_wait_range")

    def decode_walk_wait(self):
        self._check(lib().himg_hip_decode_walk_wait(self._ctx), "decode_walk_wait")

    # row-sharded encode (see himg_amd/sharded.py) --------------------------------------
    def shard_stats(self, d_frame_base, width, height, pixel_stride, channels, quality, use_ycbcr,
                    row0, row1, d_hist, d_low_rows, stream=0):
        rc = lib().himg_hip_shard_stats(self._ctx, _ptr(d_frame_base), width, height, pixel_stride,
                                        channels, quality, 1 if use_ycbcr else 0, row0, row1,
                                        _ptr(d_hist), _ptr(d_low_rows), C.c_void_p(stream))
        self._check(rc, "shard_stats")

    def shard_row_bits(self, d_hist_global, d_row_bits, stream=0):
        rc = lib().himg_hip_shard_row_bits(self._ctx, _ptr(d_hist_global), _ptr(d_row_bits),
                                           C.c_void_p(stream))
        self._check(rc, "shard_row_bits")

    def shard_emit(self, d_all_row_bits, d_rel, rel_cap, d_rel_size, stream=0):
        rc = lib().himg_hip_shard_emit(self._ctx, _ptr(d_all_row_bits), _ptr(d_rel), rel_cap,
                                       _ptr(d_rel_size), C.c_void_p(stream))
        self._check(rc, "shard_emit")

    def shard_assemble(self, d_low_full, d_all_row_bits, d_rel, rel_bytes, d_out, out_cap, d_size,
                       d_status, stream=0):
        rc = lib().himg_hip_shard_assemble(self._ctx, _ptr(d_low_full), _ptr(d_all_row_bits),
                                           _ptr(d_rel), rel_bytes, _ptr(d_out), out_cap,
                                           _ptr(d_size), _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "shard_assemble")

    def shard_head(self, d_low_full, d_all_row_bits, d_out, out_cap, d_size, d_head, d_status, stream=0):
        rc = lib().himg_hip_shard_head(self._ctx, _ptr(d_low_full), _ptr(d_all_row_bits), _ptr(d_out), out_cap,
                                       _ptr(d_size), _ptr(d_head), _ptr(d_status), C.c_void_p(stream))
        self._check(rc, "shard_head")

    def shard_finish(self, d_out, out_cap, d_size, stream=0):
        rc = lib().himg_hip_shard_finish(self._ctx, _ptr(d_out), out_cap, _ptr(d_size), C.c_void_p(stream))
        self._check(rc, "shard_finish")

    # introspection ---------------------------------------------------------------
    def debug_read(self, what, frame, nbytes, dtype=np.uint8, decoder=False):
        buf = np.empty(nbytes, np.uint8)
        n = C.c_size_t()
        sel = DBG[what] | (DBG_DECODER if decoder else 0)
        rc = lib().himg_hip_debug_read(self._ctx, sel, frame, buf.ctypes.data, nbytes, C.byref(n))
        self._check(rc, "debug_read(%s)" % what)
        return buf[: n.value].view(dtype)

    def profile(self, enable):
        lib().himg_hip_profile_enable(self._ctx, 1 if enable else 0)

    def profile_reset(self):
        lib().himg_hip_profile_reset(self._ctx)

    def profile_read(self):
        n = C.c_int()
        names = (C.c_char_p * 32)()
        ms = (C.c_double * 32)()
        cnt = (C.c_int * 32)()
        rc = lib().himg_hip_profile_read(self._ctx, C.byref(n), names, ms, cnt)
        self._check(rc, "profile_read")
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n.value)}


class MultiEngine:
    """himg_hip_create_multi: several device slots behind one handle (the same device may
    be named more than once).  encode() / decode() shard ONE frame by block rows over the
    slots; encode_batch() / decode_batch() deal independent frames over them."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._m = C.c_void_p()
        rc = lib().himg_hip_create_multi(devs, len(devices), C.byref(self._m))
        if rc:
            raise HimgError(rc, "himg_hip_create_multi (no usable GPU? there is no CPU fallback)")
        self.devices = list(devices)

    def close(self):
        if self._m:
            lib().himg_hip_destroy_multi(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise HimgError(rc, "%s: %s" % (what, lib().himg_hip_multi_last_error(self._m).decode()))

    def set_option(self, option, value):
        opt = {"fix_t2": 1}[option] if isinstance(option, str) else int(option)
        self._check(lib().himg_hip_multi_set_option(self._m, opt, int(value)), "set_option")

    def encode(self, img, quality=50, use_ycbcr=True):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        ch = img.shape[2] if img.ndim == 3 else 1
        out, n = C.c_void_p(), C.c_size_t()
        rc = lib().himg_hip_multi_encode(self._m, img.ctypes.data, w, h, ch, ch, quality, 1 if use_ycbcr else 0,
                                         C.byref(out), C.byref(n))
        self._check(rc, "multi_encode")
        a = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), (n.value,)).copy()
        lib().himg_hip_free(out)
        return a

    def decode(self, packed):
        packed = np.ascontiguousarray(np.frombuffer(packed, np.uint8) if isinstance(packed, (bytes, bytearray)) else packed)
        out = C.c_void_p()
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        rc = lib().himg_hip_multi_decode(self._m, packed.ctypes.data, packed.nbytes, C.byref(out), C.byref(w),
                                         C.byref(h), C.byref(c))
        self._check(rc, "multi_decode")
        n = w.value * h.value * c.value
        a = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), (n,)).copy().reshape(h.value, w.value, c.value)
        lib().himg_hip_free(out)
        return a

    def encode_batch(self, frames, quality=50, use_ycbcr=True):
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        n = len(frames)
        h, w = frames[0].shape[:2]
        ch = frames[0].shape[2] if frames[0].ndim == 3 else 1
        cap = max_packed_size(w, h, ch)
        outs = [np.empty(cap, np.uint8) for _ in range(n)]
        src = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
        sizes = (C.c_size_t * n)()
        rc = lib().himg_hip_multi_encode_batch(self._m, src, n, w, h, ch, ch, quality, 1 if use_ycbcr else 0,
                                               dst, caps, sizes)
        self._check(rc, "multi_encode_batch")
        return [o[: sizes[i]] for i, o in enumerate(outs)]

    def decode_batch(self, streams):
        streams = [np.ascontiguousarray(s, np.uint8) for s in streams]
        n = len(streams)
        outs = []
        for s_ in streams:
            w, h, c = C.c_int(), C.c_int(), C.c_int()
            ok = lib().himg_hip_peek(s_.ctypes.data, s_.nbytes, C.byref(w), C.byref(h), C.byref(c)) == HIMG_OK
            outs.append(np.empty(w.value * h.value * c.value if ok else 1, np.uint8))
        src = (C.c_void_p * n)(*[s_.ctypes.data for s_ in streams])
        szs = (C.c_size_t * n)(*[s_.nbytes for s_ in streams])
        dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
        ws, hs, cs = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
        rc = lib().himg_hip_multi_decode_batch(self._m, src, szs, n, dst, caps, ws, hs, cs)
        self._check(rc, "multi_decode_batch")
        return [o[: ws[i] * hs[i] * cs[i]].reshape(hs[i], ws[i], cs[i]) for i, o in enumerate(outs)]


def index_host(packed, fix_t2=False):
    """Row index of a stream in host memory (no GPU): (width, height, channels, offsets
    uint32[rows], lengths uint32[rows], rows_first).  Raises HimgError on a stream whose
    container or row headers the decoder would reject."""
    a = np.ascontiguousarray(packed, np.uint8)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib().himg_hip_peek(a.ctypes.data, a.size, C.byref(w), C.byref(h), C.byref(c))
    if rc != 0:
        raise HimgError(rc, "index_host: not a HIMG stream")
    rows = (h.value + 7) // 8
    idx = np.zeros(2 * rows, np.uint32)
    first = C.c_uint32()
    rc = lib().himg_hip_index_host(a.ctypes.data, a.size, 1 if fix_t2 else 0, C.byref(w), C.byref(h),
                                   C.byref(c), idx.ctypes.data, rows, C.byref(first))
    if rc != 0:
        raise HimgError(rc, "index_host")
    return w.value, h.value, c.value, idx[:rows], idx[rows:], first.value


def _ptr(x):
    """Device pointer of a torch tensor, or a raw integer address."""
    if x is None:
        return C.c_void_p(0)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))
