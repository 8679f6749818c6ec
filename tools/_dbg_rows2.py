import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import himg_amd
from himg_amd import sharded
W = H = 16384
img = himg_amd.synth("randtile", 0, W, H)
eng = himg_amd.Engine(0)
d = torch.from_numpy(img).to("cuda:0")
back = sharded.EngineBackend(eng, d, 0, W, H, 50, True)
out = sharded.encode_sharded(back, H // 8, W // 8, 4, True, host=False)
n = out.numel()
buf = torch.zeros((n + 15) // 16 * 16 + 64, dtype=torch.uint8, device="cuda:0"); buf[:n] = out
rows = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
st = torch.zeros(2, dtype=torch.int32, device="cuda:0")
cur = torch.cuda.current_stream().cuda_stream
ts = []
for i in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.decode_rows_device(buf, n, W, H, 4, 0, H // 8, rows, st, cur)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("HIMG_WALK_SEGS", os.environ.get("HIMG_WALK_SEGS"), "decode ms", ["%.2f" % x for x in ts], "status", int(st[0]), himg_amd.fnv1a64(rows.cpu().numpy()))
ts = []
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    ok, _ = sharded.decode_sharded(eng, out, W, H, 4, gather=False, device="cuda:0")
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("  via decode_sharded", ["%.2f" % x for x in ts])
