"""chimg / dhimg command line tools: the reference's command
line, messages and exit codes (src/chimg.cpp:36-169, src/dhimg.cpp:17-72) on top of
the drop-in Encoder / Decoder classes, with Netpbm instead of FreeImage for file I/O."""
import os
import subprocess

import numpy as np
import pytest

import himg_amd
from himg_amd import build as hb

import oracle_lib as ol


@pytest.fixture(scope="module")
def tools():
    chimg, dhimg = hb.build_cli()[:2]
    return chimg, dhimg


def _run(*cmd):
    return subprocess.run(list(cmd), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _write_pnm(path, img):
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else img.shape[2]
    with open(path, "wb") as f:
        if c == 1:
            f.write(b"P5\n# a comment\n%d %d\n255\n" % (w, h))
        elif c == 3:
            f.write(b"P6\n%d %d\n255\n" % (w, h))
        else:
            f.write(b"P7\nWIDTH %d\nHEIGHT %d\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n" % (w, h))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def _read_pnm(path):
    data = open(path, "rb").read()
    if data[:2] in (b"P5", b"P6"):
        c = 1 if data[:2] == b"P5" else 3
        # magic, width, height, maxval, then exactly one whitespace byte before the samples
        pos, toks = 0, []
        while len(toks) < 4:
            while data[pos:pos + 1].isspace():
                pos += 1
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            toks.append(data[pos:end])
            pos = end
        w, h = int(toks[1]), int(toks[2])
        return np.frombuffer(data[pos + 1: pos + 1 + w * h * c], np.uint8).reshape(h, w, c)
    head, rest = data.split(b"ENDHDR\n", 1)
    kv = dict(l.split(None, 1) for l in head.decode().splitlines()[1:] if l.strip())
    w, h, c = int(kv["WIDTH"]), int(kv["HEIGHT"]), int(kv["DEPTH"])
    return np.frombuffer(rest[: w * h * c], np.uint8).reshape(h, w, c)


def _freeimage_order(img):
    """Top-down RGB(A) -> FreeImage's bottom-up BGR(A) (what the reference chimg hands to the codec)."""
    img = img if img.ndim == 3 else img[:, :, None]
    out = img[::-1].copy()
    if img.shape[2] >= 3:
        out[:, :, [0, 2]] = out[:, :, [2, 0]]
    return np.ascontiguousarray(out)


def test_usage_and_argument_errors(tools, tmp_path):
    chimg, dhimg = tools
    r = _run(chimg)
    assert r.returncode == 0 and r.stdout.startswith("Usage: %s [options] image outfile\nOptions:\n" % chimg)
    assert " -q <quality> Set the quality (0-100)\n -rgb         Use RGB color space (instead of YCbCr)\n" in r.stdout
    r = _run(chimg, "-q", "101", "a", "b")
    assert r.returncode == 0 and r.stdout.startswith("Invalid quality level: 101\nUsage:")
    r = _run(chimg, "-q", "x7", "a", "b")
    assert r.returncode == 0 and r.stdout.startswith("Invalid integer expression: x7\nUsage:")
    r = _run(chimg, "-zz", "a", "b")
    assert r.returncode == 0 and r.stdout.startswith("Invalid option: -zz\nUsage:")
    r = _run(dhimg, "only-one")
    assert r.returncode == 0 and r.stdout == "Usage: %s image outfile\n" % dhimg
    r = _run(dhimg, str(tmp_path / "missing.himg"), str(tmp_path / "o.ppm"))
    assert r.returncode == 255 and r.stdout == "Unable to read file %s\n" % (tmp_path / "missing.himg")
    # input files chimg cannot use (exit -1, message on stderr like the reference)
    r = _run(chimg, str(tmp_path / "missing.ppm"), str(tmp_path / "o.himg"))
    assert r.returncode == 255 and r.stderr == "Unable to load %s\n" % (tmp_path / "missing.ppm")
    bad = tmp_path / "bad.ppm"
    bad.write_bytes(b"GIF89a....")
    r = _run(chimg, str(bad), str(tmp_path / "o.himg"))
    assert r.returncode == 255 and r.stderr == "Unknown file format for %s\n" % bad


@pytest.mark.gpu
@pytest.mark.parametrize("channels,flags", [(3, []), (4, ["-q", "70"]), (1, ["-q", "90"]), (3, ["-rgb", "-q", "30"])])
def test_cli_round_trip_matches_the_oracle(tools, tmp_path, channels, flags):
    chimg, dhimg = tools
    w, h = 256, 128
    rgba = himg_amd.synth("randtile", 11 + channels, w, h)
    img = rgba[:, :, 0] if channels == 1 else rgba[:, :, :channels]
    src = str(tmp_path / "in.pnm")
    _write_pnm(src, img)
    packed_path, out_path = str(tmp_path / "o.himg"), str(tmp_path / "o.pnm")
    r = _run(chimg, *flags, src, packed_path)
    q = int(flags[flags.index("-q") + 1]) if "-q" in flags else 50
    want = ol.oracle_encode(_freeimage_order(img), q, "-rgb" not in flags, channels=channels, stride=channels)
    assert r.returncode == 0, r.stderr
    # the library's two lines (encoder.cpp:219,334), then the tool's own
    assert r.stdout.splitlines()[-1] == "Compressed size: %d" % len(want)
    assert r.stdout.startswith("Low resolution data: ")
    got = np.fromfile(packed_path, np.uint8)
    assert np.array_equal(got, want)

    r = _run(dhimg, packed_path, out_path)
    rc, pix = ol.oracle_decode(want)
    if rc != 0:   # the reference refuses its own stream (trap T2)
        assert r.returncode == 255 and r.stdout.splitlines()[-1] == "Unable to decode image."
        return
    assert r.returncode == 0, r.stdout
    assert r.stdout.splitlines()[0] == "File size: %d" % len(want)
    back = _read_pnm(out_path)
    assert np.array_equal(_freeimage_order(back), pix.reshape(h, w, channels))


@pytest.mark.gpu
def test_dhimg_fixed_mode_through_the_environment(tools, tmp_path):
    """HIMG_FIX_T2=1 switches the opt-in fixed mode on for callers that only see the
    C++ classes: a flat picture, which the reference cannot decode (trap T2)."""
    chimg, dhimg = tools
    img = np.full((64, 96, 3), 90, np.uint8)
    src, packed, out = str(tmp_path / "flat.ppm"), str(tmp_path / "flat.himg"), str(tmp_path / "out.ppm")
    _write_pnm(src, img)
    assert _run(chimg, src, packed).returncode == 0
    r = _run(dhimg, packed, out)
    assert r.returncode == 255 and r.stdout.splitlines()[-1] == "Unable to decode image."
    r = subprocess.run([dhimg, packed, out], env=dict(os.environ, HIMG_FIX_T2="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout
    back = _read_pnm(out)
    assert back.shape == img.shape and np.abs(back.astype(int) - img.astype(int)).max() <= 2


def _benchmark():
    return hb.build_cli()[2]


def test_benchmark_usage():
    exe = _benchmark()
    r = _run(exe)
    assert r.returncode == 0 and r.stdout == "Usage: %s [-d][-e] image\n  -d Decode (default)\n  -e Encode\n" % exe
    r = _run(exe, "a", "b")
    assert r.returncode == 0 and r.stdout.startswith("Usage: ")


@pytest.mark.gpu
def test_benchmark_decode_and_encode(tools, tmp_path):
    """The reference's timing protocol (benchmark.cpp:108-156): one Decoder, 30
    decodes, iteration lines, then Min / Max / Average."""
    chimg, _ = tools
    exe = _benchmark()
    img = himg_amd.synth("randtile", 2, 512, 256)[:, :, :3]
    src, packed = str(tmp_path / "in.ppm"), str(tmp_path / "in.himg")
    _write_pnm(src, img)
    assert _run(chimg, "-q", "80", src, packed).returncode == 0
    for args in ([packed], ["-d", packed], ["-e", src]):
        r = _run(exe, *args)
        lines = r.stdout.splitlines()
        assert r.returncode == 0, r.stdout
        assert lines[0].startswith("File size: ")
        assert lines[1:31] == ["Iteration %d/30" % i for i in range(1, 31)]
        assert lines[31].startswith("    Min: ") and lines[32].startswith("    Max: ") and lines[33].startswith("Average: ")
        assert float(lines[33].split()[1]) > 0
    # a stream the reference cannot decode (flat picture, trap T2)
    flat = str(tmp_path / "flat.ppm")
    _write_pnm(flat, np.full((64, 64, 3), 9, np.uint8))
    assert _run(chimg, flat, packed).returncode == 0
    r = _run(exe, packed)
    assert r.returncode == 255 and r.stdout.splitlines()[-1] == "Unable to decode image."
