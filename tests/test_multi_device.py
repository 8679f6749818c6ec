"""Several GPUs behind the C ABI (include/himg_hip.h, "multi-device"):
himg_hip_create_multi over device slots -- here slots that all map to GPU 0, the test box
has one GPU -- one frame sharded by block rows, batches dealt over the slots, and the
drop-in C++ classes picking the handle up through HIMG_DEVICES."""
import os
import subprocess

import numpy as np
import pytest

import himg_amd
from himg_amd import build as hb

import oracle_lib as ol
from test_cli import _write_pnm, _read_pnm, _freeimage_order


def test_create_multi_fails_loudly_without_a_gpu():
    """No CPU fallback: on a machine without a usable GPU the handle is not created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(himg_amd.HimgError) as e:
        himg_amd.MultiEngine([0])
    assert e.value.code == himg_amd.HIMG_ERR_HIP
    L = himg_amd.lib()
    assert L.himg_hip_multi_count(None) == 0
    assert L.himg_hip_create_multi(None, 1, None) == himg_amd.HIMG_ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("slots,staged", [(1, 0), (2, 0), (2, 1), (3, 0), (8, 0)])
@pytest.mark.parametrize("kind,w,h,q", [("randtile", 512, 1024, 50), ("gradn", 256, 520, 90)])
def test_multi_one_frame_matches_the_oracle(slots, staged, kind, w, h, q, monkeypatch):
    """One frame, block rows sharded over `slots` device slots: the stream is the
    oracle's byte for byte (direct peer writes and the staged peer-copy form), the
    decoded picture is the oracle's."""
    monkeypatch.setenv("HIMG_MULTI_STAGED", str(staged))
    img = himg_amd.synth(kind, 3, w, h)
    want = ol.oracle_encode(img, q, True)
    rc, pix = ol.oracle_decode(want)
    assert rc == 0
    m = himg_amd.MultiEngine([0] * slots)
    assert himg_amd.lib().himg_hip_multi_count(m._m) == slots
    for _ in range(2):     # the handle is reusable
        got = m.encode(img, q, True)
        assert np.array_equal(got, want)
        assert np.array_equal(m.decode(want), pix)
    m.close()


@pytest.mark.gpu
def test_multi_rejects_like_the_reference():
    """Trap T2: a stream the reference decoder refuses is refused (HIMG_ERR_FORMAT), and
    a damaged row header as well; the fixed mode decodes the T2 stream on every slot."""
    packed = ol.oracle_encode(himg_amd.synth("grad", 0, 512, 512), 50, True)
    assert ol.oracle_decode(packed)[0] != 0
    m = himg_amd.MultiEngine([0, 0])
    with pytest.raises(himg_amd.HimgError) as e:
        m.decode(packed)
    assert e.value.code == himg_amd.HIMG_ERR_FORMAT
    good = ol.oracle_encode(himg_amd.synth("randtile", 0, 512, 512), 50, True)
    _, _, _, off, ln, first = himg_amd.index_host(good)
    bad = good.copy()
    bad[first + 1] |= 0x7f
    assert ol.oracle_decode(bad)[0] != 0
    with pytest.raises(himg_amd.HimgError):
        m.decode(bad)
    m.set_option("fix_t2", 1)
    rc, pix = ol.oracle_decode(packed, fix_t2=True)
    assert rc == 0 and np.array_equal(m.decode(packed), pix)
    m.close()


@pytest.mark.gpu
def test_multi_batches_are_dealt_over_the_slots():
    frames = [himg_amd.synth("randtile", s, 256, 128) for s in range(7)]
    want = [ol.oracle_encode(f, 50, True) for f in frames]
    m = himg_amd.MultiEngine([0, 0, 0])
    got = m.encode_batch(frames, 50, True)
    assert all(np.array_equal(g, w_) for g, w_ in zip(got, want))
    pix = m.decode_batch(want)
    for p, w_ in zip(pix, want):
        rc, ref = ol.oracle_decode(w_)
        assert rc == 0 and np.array_equal(p, ref)
    m.close()


@pytest.mark.gpu
def test_cpp_classes_use_himg_devices(tmp_path):
    """The reference's command line through the drop-in classes with HIMG_DEVICES naming two
    slots: himg::Encoder / himg::Decoder shard the frame, files and messages stay the same."""
    chimg, dhimg = hb.build_cli()[:2]
    img = himg_amd.synth("randtile", 5, 512, 1024)
    src, packed_path, out_path = tmp_path / "in.pam", tmp_path / "out.himg", tmp_path / "back.pam"
    _write_pnm(src, img)
    env = dict(os.environ, HIMG_DEVICES="0,0")
    r = subprocess.run([chimg, "-q", "70", str(src), str(packed_path)], stdout=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stdout
    want = ol.oracle_encode(_freeimage_order(img), 70, True)
    assert np.array_equal(np.fromfile(packed_path, np.uint8), want)
    assert "Low resolution data: " in r.stdout and "Compressed size: %d" % want.size in r.stdout
    r = subprocess.run([dhimg, str(packed_path), str(out_path)], stdout=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stdout
    rc, pix = ol.oracle_decode(want)
    assert np.array_equal(_freeimage_order(_read_pnm(out_path)), pix.reshape(1024, 512, 4))
