"""GPU (-m gpu): the forms of the decoder's row kernel chosen by environment knobs
(HIMG_PERSIST_ROWS: persistent workgroups or one workgroup per row; HIMG_PREFETCH_ROWS: touch
loads for the next row's packed bytes) decode the same streams to the same pixels as the CPU
oracle.  The knobs are read once per process, so every form runs in a child process."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
    import himg_amd
    import oracle_lib as ol
    eng = himg_amd.Engine(0)
    # (width, height, frames): more grid elements than the GPU has CUs, so that a persistent
    # workgroup takes several; 4096 px (one row per workgroup, compile-time strides), 1920 px
    # (two rows per workgroup), 1000 px (run-time strides, ragged tiles), 3 channels (general form).
    for w, h, B, ch in ((4096, 64, 40, 4), (1920, 136, 40, 4), (1000, 72, 96, 4), (520, 64, 96, 3)):
        frames = [himg_amd.synth("randtile", s, w, h)[:, :, :ch].copy() for s in range(3)]
        streams = [eng.encode(f, 50, True, channels=ch, pixel_stride=ch) for f in frames]
        want = []
        for s in streams:
            rc, px = ol.oracle_decode(s)
            assert rc == 0
            want.append(px)
        cap = (max(len(s) for s in streams) + 255) // 256 * 256
        d_in = torch.zeros((B, cap), dtype=torch.uint8, device="cuda")
        sizes = np.zeros(B, np.uint32)
        for b in range(B):
            s = streams[b %% 3]
            d_in[b, : len(s)] = torch.from_numpy(np.asarray(s)).cuda()
            sizes[b] = len(s)
        d_pix = torch.empty((B, h, w, ch), dtype=torch.uint8, device="cuda")
        d_st = torch.ones(B, dtype=torch.int32, device="cuda")
        eng.decode_device(d_in, cap, sizes, B, w, h, ch, d_pix, d_st, 0)
        torch.cuda.synchronize()
        assert not d_st.cpu().numpy().any(), (w, h, "status")
        pix = d_pix.cpu().numpy()
        for b in range(B):
            assert np.array_equal(pix[b], want[b %% 3].reshape(h, w, ch)), (w, h, b)
    print("forms ok")
""")


@pytest.mark.parametrize("persist,prefetch", [("1", "1"), ("0", "1"), ("1", "0"), ("0", "0"), ("7", "1")])
def test_row_kernel_forms_match_oracle(persist, prefetch):
    env = dict(os.environ, HIMG_PERSIST_ROWS=persist, HIMG_PREFETCH_ROWS=prefetch)
    code = CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forms ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
