"""ctypes access to the CPU checker: our C restatement (oracle/libhimg_oracle.so)
and, when it has been built, the REAL reference (oracle/_ref/libhimg_ref.so).

Test infrastructure only -- nothing under himg_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libhimg_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libhimg_ref.so")


class Trace(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("channels", C.c_int), ("rows", C.c_int),
        ("cols", C.c_int), ("use_ycbcr", C.c_int),
        ("lifted", C.POINTER(C.c_uint8)), ("avg", C.POINTER(C.c_uint8)),
        ("lowres", C.POINTER(C.c_uint8)), ("lres_sym", C.POINTER(C.c_uint8)),
        ("lres_sym_size", C.c_int), ("fres_sym", C.POINTER(C.c_uint8)),
        ("fres_sym_size", C.c_int),
        ("lres_hist", C.c_uint32 * 261), ("fres_hist", C.c_uint32 * 261),
        ("lres_len", C.c_uint8 * 261), ("fres_len", C.c_uint8 * 261),
        ("lres_code", C.c_uint64 * 261), ("fres_code", C.c_uint64 * 261),
        ("lres_tree_bytes", C.c_int), ("fres_tree_bytes", C.c_int),
        ("fres_row_bytes", C.POINTER(C.c_int)),
        ("shift_luma", C.c_uint8 * 64), ("shift_chroma", C.c_uint8 * 64),
        ("lmap", C.c_int16 * 128), ("fmap", C.c_int16 * 128),
    ]


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.run(["make", "-s", "-C", ORACLE_DIR, "libhimg_oracle.so"], check=True)
        L = C.CDLL(ORACLE_SO)
        L.himg_oracle_map_to_8bit.restype = C.c_uint8
        L.himg_oracle_free.restype = None
        L.himg_oracle_trace_free.restype = None
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.himg_ref_free.restype = None
    return _ref


def _take(ptr, n, free):
    a = np.ctypeslib.as_array(ptr, (n,)).copy() if n else np.zeros(0, np.uint8)
    free(ptr)
    return a


def oracle_encode(img, quality=50, use_ycbcr=True, channels=None, stride=None, trace=False):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = channels if channels is not None else (img.shape[2] if img.ndim == 3 else 1)
    st = stride if stride is not None else (img.shape[2] if img.ndim == 3 else 1)
    out, n = C.POINTER(C.c_uint8)(), C.c_int()
    tr = Trace() if trace else None
    rc = oracle().himg_oracle_encode(img.ctypes.data_as(C.c_void_p), w, h, st, ch, quality,
                                     1 if use_ycbcr else 0, C.byref(out), C.byref(n),
                                     C.byref(tr) if trace else None)
    assert rc == 0, rc
    packed = _take(out, n.value, oracle().himg_oracle_free)
    if not trace:
        return packed
    rows, cols = tr.rows, tr.cols
    t = {
        "rows": rows, "cols": cols,
        "avg": np.ctypeslib.as_array(tr.avg, (ch * rows * cols,)).copy(),
        "lowres": np.ctypeslib.as_array(tr.lowres, (ch * rows * cols,)).copy(),
        "lres_sym": np.ctypeslib.as_array(tr.lres_sym, (tr.lres_sym_size,)).copy(),
        "fres_sym": np.ctypeslib.as_array(tr.fres_sym, (tr.fres_sym_size,)).copy(),
        "lres_hist": np.array(tr.lres_hist, np.uint32), "fres_hist": np.array(tr.fres_hist, np.uint32),
        "lres_len": np.array(tr.lres_len, np.uint32), "fres_len": np.array(tr.fres_len, np.uint32),
        "lres_code": np.array(tr.lres_code, np.uint64), "fres_code": np.array(tr.fres_code, np.uint64),
        "lres_tree_bytes": tr.lres_tree_bytes, "fres_tree_bytes": tr.fres_tree_bytes,
        "fres_row_bytes": np.ctypeslib.as_array(tr.fres_row_bytes, (rows,)).astype(np.uint32).copy(),
        "shift_luma": np.array(tr.shift_luma, np.uint8), "shift_chroma": np.array(tr.shift_chroma, np.uint8),
        "lmap": np.array(tr.lmap, np.int16), "fmap": np.array(tr.fmap, np.int16),
    }
    oracle().himg_oracle_trace_free(C.byref(tr))
    return packed, t


def oracle_decode(packed, threads=1, fix_t2=False):
    """fix_t2: the oracle's test knob for the product's opt-in fixed mode (trap T2);
    the default is the reference's behaviour."""
    packed = np.ascontiguousarray(packed, np.uint8)
    oracle().himg_oracle_set_compat_fix(1 if fix_t2 else 0)
    try:
        return _oracle_decode(packed, threads)
    finally:
        oracle().himg_oracle_set_compat_fix(0)


def _oracle_decode(packed, threads):
    out = C.POINTER(C.c_uint8)()
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = oracle().himg_oracle_decode(packed.ctypes.data_as(C.c_void_p), packed.nbytes, threads,
                                     C.byref(out), C.byref(w), C.byref(h), C.byref(c))
    if rc != 0:
        return rc, None
    n = w.value * h.value * c.value
    return 0, _take(out, n, oracle().himg_oracle_free).reshape(h.value, w.value, c.value)


def oracle_decode_trace(packed):
    packed = np.ascontiguousarray(packed, np.uint8)
    out, ls, fs, low = (C.POINTER(C.c_uint8)() for _ in range(4))
    w, h, c, ln, fn = (C.c_int() for _ in range(5))
    rc = oracle().himg_oracle_decode_trace(packed.ctypes.data_as(C.c_void_p), packed.nbytes,
                                           C.byref(out), C.byref(w), C.byref(h), C.byref(c),
                                           C.byref(ls), C.byref(ln), C.byref(fs), C.byref(fn),
                                           C.byref(low))
    if rc != 0:
        return rc, None
    fr = oracle().himg_oracle_free
    rows, cols = (h.value + 7) // 8, (w.value + 7) // 8
    res = {
        "pixels": _take(out, w.value * h.value * c.value, fr).reshape(h.value, w.value, c.value),
        "lres_sym": _take(ls, ln.value, fr), "fres_sym": _take(fs, fn.value, fr),
        "lowres": _take(low, c.value * rows * cols, fr),
    }
    return 0, res


def ref_encode(img, quality=50, use_ycbcr=True, channels=None, stride=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = channels if channels is not None else (img.shape[2] if img.ndim == 3 else 1)
    st = stride if stride is not None else (img.shape[2] if img.ndim == 3 else 1)
    out, n = C.POINTER(C.c_uint8)(), C.c_int()
    rc = ref().himg_ref_encode(img.ctypes.data_as(C.c_void_p), w, h, st, ch, quality,
                               1 if use_ycbcr else 0, C.byref(out), C.byref(n))
    assert rc == 0
    return _take(out, n.value, ref().himg_ref_free)


def ref_decode(packed, threads=1):
    packed = np.ascontiguousarray(packed, np.uint8)
    out = C.POINTER(C.c_uint8)()
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = ref().himg_ref_decode(packed.ctypes.data_as(C.c_void_p), packed.nbytes, threads,
                               C.byref(out), C.byref(w), C.byref(h), C.byref(c))
    if rc != 0:
        return rc, None
    n = w.value * h.value * c.value
    return 0, _take(out, n, ref().himg_ref_free).reshape(h.value, w.value, c.value)
