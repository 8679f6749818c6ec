#!/usr/bin/env python3
"""Per-kernel times of the row-sharded encode / decode of one large frame on one GPU
(args: width height [ranks simulated = 1]) -- profiles/rNN_cfg4_rows.json (GPU box)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import himg_amd
from himg_amd import sharded
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
img = himg_amd.synth("randtile", 0, W, H)
rows, cols = H // 8, W // 8
eng = himg_amd.Engine(0)
d = torch.from_numpy(img).to("cuda:0")
back = sharded.EngineBackend(eng, d, 0, W, H, 50, True)
out = sharded.encode_sharded(back, rows, cols, 4, True, host=False)
res = {"width": W, "height": H, "packed_size": int(out.numel()), "stream_fnv": himg_amd.fnv1a64(out.cpu().numpy())}
for what in ("encode", "decode"):
    eng.profile_reset(); eng.profile(True)
    n = 5
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(n):
        if what == "encode":
            sharded.encode_sharded(back, rows, cols, 4, True, host=False)
        else:
            ok, _ = sharded.decode_sharded(eng, out, W, H, 4, gather=False, device="cuda:0")
            assert ok
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n * 1e3
    eng.profile(False)
    # per frame (n profiled frames): a kernel launched twice per frame counts with both launches
    st = {k: round(v[0] / n, 4) for k, v in sorted(eng.profile_read().items(), key=lambda kv: -kv[1][0])}
    res[what] = {"ms_per_frame": round(dt, 3), "stages_ms": st}
print(json.dumps(res, indent=1))
