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
# mode 1: every coefficient at or one beyond the largest magnitude the packed path's
# range conditions allow (packed_wht_exact), with the sign patterns that maximise the
# butterfly sums.
# variant: 0 the generic gather, -1 the identity-range gather (codes of rows 2..7 x columns
# 1..7 taken as they are when a wavefront's all lie in the identity range of the companding
# table: amplitude 1 makes the table the identity, so planes of small codes take it and
# planes with larger ones fall back) with the run-time tile count, 64 the same with the
# count at compile time.
@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, -1, 64])
@pytest.mark.parametrize("amp,zeros,mode", [(1, 60, 0), (1, 60, 2), (1, 95, 2), (2, 90, 0), (4, 60, 0), (16, 30, 0), (300, 60, 0), (1, 0, 1)])
def test_tile_plane_matches_the_scalar_model(tile_plane_check, amp, zeros, mode, variant):
    """tile_plane (kernels_dec.hip) == the int32 arithmetic of hadamard.cpp:47-74,
    quantize.cpp:153-165, downsampled.cpp:116-169 and decoder.cpp:401-413 on
    16384 random planes."""
    r = subprocess.run([tile_plane_check, str(amp), str(zeros), str(mode), str(variant)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatching pixels 0" in r.stdout
    assert "lowres_quads mismatches 0" in r.stdout
