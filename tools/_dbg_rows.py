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
print("fnv", himg_amd.fnv1a64(out.cpu().numpy()))
def run(label, **kw):
    ts = []
    for i in range(6):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ok, _ = sharded.decode_sharded(eng, out, W, H, 4, gather=False, device="cuda:0", **kw)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
        assert ok
    print(label, ["%.2f" % x for x in ts])
run("engine stream (default)")
cur = torch.cuda.current_stream().cuda_stream
run("caller's stream = torch current", stream=cur)
# plain row-range decode on the current stream (what world == 1 did before)
dec = list(eng._sharded_decoders.values())[0]
st = torch.zeros(2, dtype=torch.int32, device="cuda:0")
buf = dec._buffer(out.numel()); buf[:out.numel()] = out
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.decode_rows_device(buf, out.numel(), W, H, 4, 0, H // 8, dec.d_rows, st, cur)
    torch.cuda.synchronize(); print("direct, current stream %.2f ms" % ((time.perf_counter() - t) * 1e3))
es = torch.cuda.Stream()
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.decode_rows_device(buf, out.numel(), W, H, 4, 0, H // 8, dec.d_rows, st, es.cuda_stream)
    torch.cuda.synchronize(); print("direct, side torch stream %.2f ms" % ((time.perf_counter() - t) * 1e3))
for cw in (0, 1, -1):
    eng.set_option("count_wave", cw)
    eng.profile_reset(); eng.profile(True)
    for i in range(3):
        eng.decode_rows_device(buf, out.numel(), W, H, 4, 0, H // 8, dec.d_rows, st, cur)
    torch.cuda.synchronize()
    eng.profile(False)
    print("count_wave", cw, {k: round(v[0] / 3, 3) for k, v in sorted(eng.profile_read().items(), key=lambda kv: -kv[1][0])[:6]}, int(st[0]))
