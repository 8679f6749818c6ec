"""Device-side unit checks built by himg_amd.build.build_unit_checks()."""
import subprocess

import pytest

from himg_amd import build as hb


@pytest.fixture(scope="module")
def tile_plane_check():
    return hb.build_unit_checks()[0]


# amplitude of the dequantised coefficients, percentage of zero codes: 1-2 keep
# every plane on the packed int16 path, 4 mixes both paths inside a wave, 16 and
# 300 put (nearly) every plane on the scalar int32 path (int16 wrap included).
@pytest.mark.gpu
@pytest.mark.parametrize("amp,zeros", [(1, 60), (2, 90), (4, 60), (16, 30), (300, 60)])
def test_tile_plane_matches_the_scalar_model(tile_plane_check, amp, zeros):
    """tile_plane (kernels_dec.hip) == the int32 arithmetic of hadamard.cpp:47-74,
    quantize.cpp:153-165, downsampled.cpp:116-169 and decoder.cpp:401-413 on
    16384 random planes."""
    r = subprocess.run([tile_plane_check, str(amp), str(zeros)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatching pixels 0" in r.stdout
    assert "lowres_quads mismatches 0" in r.stdout
