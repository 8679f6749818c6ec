"""CPU: host-side format tables, generators and the C-ABI surface (no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import himg_amd
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# read back from the reference's QCFG chunks.
Q50_LUMA = [4, 3, 3, 4, 5, 5, 6, 6, 4, 4, 4, 4, 5, 6, 6, 6, 4, 4, 4, 5, 5, 6, 6, 6, 4, 4, 4, 5, 6, 6,
            6, 6, 4, 4, 5, 6, 6, 7, 7, 6, 5, 5, 6, 6, 6, 7, 7, 6, 6, 6, 6, 6, 7, 7, 7, 7, 6, 6, 6, 7,
            7, 7, 7, 7]
Q50_CHROMA_HEAD = [4, 4, 5, 5, 7, 7, 7, 7, 4, 4, 5, 6, 7, 7, 7, 7, 5, 5, 6, 7, 7, 7, 7, 7, 5, 6, 7,
                   7, 7, 7, 7, 7]
Q90_LUMA_ROW0 = [1, 0, 0, 1, 2, 2, 3, 3]
LMAP_Q50 = [1, 3, 4, 5, 6, 8, 9, 10, 11, 13, 14, 15, 16, 18, 19, 20, 21, 23, 24, 25, 26, 28, 29, 30]
LMAP_Q0 = [8, 15, 23, 30, 38, 45, 53, 60, 70, 81, 97, 114, 137, 160, 190, 221, 255, 255]


def _shift(q, chroma):
    a = np.zeros(64, np.uint8)
    himg_amd.lib().himg_tables_shift(q, chroma, a.ctypes.data)
    return a


def _oracle_shift(q, chroma):
    a = np.zeros(64, np.uint8)
    ol.oracle().himg_oracle_shift_table(q, chroma, a.ctypes.data_as(C.c_void_p))
    return a


def _lmap(q, which="host"):
    a = np.zeros(128, np.int16)
    if which == "host":
        himg_amd.lib().himg_tables_lowres_map(q, a.ctypes.data)
    else:
        ol.oracle().himg_oracle_lowres_map_table(q, a.ctypes.data_as(C.c_void_p))
    return a


def test_shift_tables_known_answers():
    assert list(_shift(50, 0)) == Q50_LUMA
    assert list(_shift(50, 1)[:32]) == Q50_CHROMA_HEAD and set(_shift(50, 1)[32:]) == {7}
    assert list(_shift(90, 0)[:8]) == Q90_LUMA_ROW0
    assert list(_shift(0, 0)[:8]) == [10, 9, 9, 10, 11, 11, 12, 12] and _shift(0, 0).max() == 13
    assert not _shift(100, 0).any() and not _shift(100, 1).any()


@pytest.mark.parametrize("q", list(range(0, 101, 5)) + [1, 7, 33, 99, 255, -3])
def test_tables_match_oracle(q):
    for chroma in (0, 1):
        assert np.array_equal(_shift(q, chroma), _oracle_shift(q, chroma))
    assert np.array_equal(_lmap(q), _lmap(q, "oracle"))


def test_lowres_map_known_answers():
    assert list(_lmap(50)[1:25]) == LMAP_Q50
    assert list(_lmap(0)[1:19]) == LMAP_Q0
    assert list(_lmap(100)[1:25]) == list(range(1, 25))
    for q in (0, 10, 30, 50, 70, 90, 100):
        assert _lmap(q)[127] == 255


def test_companding_matches_oracle_everywhere():
    L = himg_amd.lib()
    fm = np.zeros(128, np.int16)
    L.himg_tables_fullres_map(fm.ctypes.data)
    assert fm[49] == 49 and fm[50] == 51 and fm[127] == 8039
    o = ol.oracle()
    for x in list(range(-600, 601)) + [-32768, -32767, -16320, 16320, 8038, 8039, 8040, 7608, 7823, 7824, 32767]:
        assert L.himg_tables_map_to_8bit(fm.ctypes.data, x) == o.himg_oracle_map_to_8bit(
            fm.ctypes.data_as(C.c_void_p), x), x
    for q in (0, 10, 50, 100):
        lm = _lmap(q)
        for x in range(-255, 256):
            assert L.himg_tables_map_to_8bit(lm.ctypes.data, x) == o.himg_oracle_map_to_8bit(
                lm.ctypes.data_as(C.c_void_p), x)
    # Quirks of Mapper::MapTo8Bit (trap T7).
    assert L.himg_tables_map_to_8bit(fm.ctypes.data, 0) == 0
    assert L.himg_tables_map_to_8bit(fm.ctypes.data, 1) == 1
    assert L.himg_tables_map_to_8bit(fm.ctypes.data, 7608) == 127      # >= table[126] -> 127
    assert L.himg_tables_map_to_8bit(fm.ctypes.data, -1) == 255


def test_hadamard_is_sequency_ordered_and_invertible():
    o = ol.oracle()
    rng = np.random.default_rng(1)
    for _ in range(20):
        x = rng.integers(-255, 256, 64).astype(np.int16)
        y = np.zeros(64, np.int16)
        z = np.zeros(64, np.int16)
        o.himg_oracle_hadamard_forward(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
        o.himg_oracle_hadamard_inverse(z.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
        assert np.array_equal(x, z)          # M*M = 8I per pass, >>3 per pass
    # Row k of the 1-D transform has exactly k sign changes.
    for k in range(8):
        e = np.zeros(64, np.int16)
        e[k] = 1   # unit impulse in row 0 -> coefficients [*, 0..7] of row 0 = column k of M
        y = np.zeros(64, np.int16)
        o.himg_oracle_hadamard_forward(y.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p))
        col = y[:8]
        assert set(np.abs(col)) == {1}
    M = np.zeros((8, 8), int)
    for k in range(8):
        e = np.zeros(64, np.int16)
        e[k] = 1
        y = np.zeros(64, np.int16)
        o.himg_oracle_hadamard_forward(y.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p))
        M[:, k] = y[:8]
    assert [int(np.sum(M[r, 1:] != M[r, :-1])) for r in range(8)] == list(range(8))
    assert np.array_equal(M @ M, 8 * np.eye(8, dtype=int))


def test_generators_match_recorded_hashes():
    assert himg_amd.fnv1a64(himg_amd.synth("grad", 0, 64, 64)) == "6a51045d4909278f"
    assert himg_amd.fnv1a64(himg_amd.synth("gradn", 0, 64, 64)) == "6a603e2aaa3a52fe"
    assert himg_amd.fnv1a64(himg_amd.synth("rand", 0, 64, 64)) == "18fdd58c1902a8cb"
    assert himg_amd.fnv1a64(himg_amd.synth("randtile", 0, 64, 64)) == "2b197bf0df513308"
    assert himg_amd.fnv1a64(himg_amd.synth("randtile", 0, 1920, 1080)) == "25c3045d0ceda4b8"


def test_abi_exports_every_declared_symbol():
    """The shared library loads and exports exactly what include/himg_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "himg_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(himg_[a-z0-9_]+)\s*\(", hdr))
    assert {"himg_hip_create", "himg_hip_encode", "himg_hip_decode", "himg_hip_encode_device",
            "himg_hip_decode_device", "himg_hip_max_packed_size", "himg_synth_fill"} <= names
    L = himg_amd.lib()
    for n in sorted(names):
        assert hasattr(L, n), "missing export: " + n


def test_max_packed_size_bounds_golden_streams():
    from golden_util import GOLDEN
    for rec in GOLDEN.values():
        cap = himg_amd.max_packed_size(rec["width"], rec["height"], rec["channels"])
        assert cap % 256 == 0 and cap >= rec["packed_size"]


def test_no_gpu_means_loud_failure():
    """There is no CPU fallback: without a device the engine refuses to start."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(himg_amd.HimgError):
        himg_amd.Engine(0)


def test_peek_reads_the_geometry_without_a_gpu():
    """himg_hip_peek parses the FRMT chunk only (decoder.cpp:144-200): no device needed."""
    import ctypes as C
    L = himg_amd.lib()
    gdir = os.path.join(ROOT, "tests", "golden")
    seen = 0
    for fn in sorted(os.listdir(gdir)):
        if not fn.endswith(".himg"):
            continue
        data = np.fromfile(os.path.join(gdir, fn), np.uint8)
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        assert L.himg_hip_peek(data.ctypes.data, data.nbytes, C.byref(w), C.byref(h), C.byref(c)) == 0
        assert (w.value, h.value) == (64, 64) and c.value in (1, 3, 4)
        seen += 1
        # truncated before the FRMT body / not a RIFF file
        assert L.himg_hip_peek(data.ctypes.data, 20, C.byref(w), C.byref(h), C.byref(c)) == himg_amd.HIMG_ERR_FORMAT
        bad = data.copy()
        bad[0] ^= 1
        assert L.himg_hip_peek(bad.ctypes.data, bad.nbytes, C.byref(w), C.byref(h), C.byref(c)) == himg_amd.HIMG_ERR_FORMAT
    assert seen >= 5
