"""Where the pair sweep's wavefronts spend their time (one cloud): GG_PAIR_TIMING=1 python tools/pair_timing.py"""
import ctypes as C, os, sys
os.environ["GG_PAIR_TIMING"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth

cloud = synth.hdl64_cloud(seed=20240113)
stride = (len(cloud) + 63) // 64 * 64
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=stride)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    seg.debug_set_tuning(k, int(v))
host = np.zeros((1, stride), dtype=api.POINT16_DTYPE); host[0, :len(cloud)] = api.pack16(cloud)
pts = torch.from_numpy(host.view(np.uint8).reshape(1, stride, 16)).cuda()
out = None
for _ in range(6):
    out = seg.filter_batch(pts, [len(cloud)], np.zeros((1, 3), np.float32), np.full(1, -1.73), out=out)
seg.synchronize()
buf = (C.c_ulonglong * 2048)()
fn = seg._L.gg_debug_pair_timing
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert fn(seg._ctx, buf) == 0
d = np.array(buf[:1024], dtype=np.uint64).reshape(32, 32).astype(np.int64)
t0 = min(int(d[w, 0]) for w in range(32) if d[w, 0])
for w in range(32):
    if not d[w, 0]:
        continue
    wg, wave = divmod(w, 16)
    line = f"wg {wg} wave {wave:2d}: start {int(d[w,0])-t0:7d} end {int(d[w,1])-t0:7d} |"
    if wave < 6:
        for gi in range(4):
            b = 2 + gi * 6
            if d[w, b]:
                line += f" group{gi}: {int(d[w,b])-t0:7d} starts done {int(d[w,b+1])-t0:7d} end {int(d[w,b+2])-t0:7d} corner wait {int(d[w,b+3]):6d} import wait {int(d[w,b+4]):6d} ({int(d[w,b+5])} waits)"
    else:
        line += " batches (records in, rings done): " + " ".join(f"{int(d[w,2+2*k])-t0}/{int(d[w,3+2*k])-t0}" for k in range(3) if d[w, 2 + 2 * k])
    print(line)
