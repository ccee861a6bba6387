"""Experiment: two contexts with half the clouds each, on two HIP streams -- the latency-bound kernels (k_reduce, k_sweep,
k_patch) of one half overlap the bandwidth-bound ones (k_classify, k_scatter, k_label) of the other.  One context orders its
batches across streams with events (ABI v2), so overlap needs two contexts.  usage: python tools/two_stream.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from groundgrid_amd import api
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
clouds = bench.make_clouds(B, 0)
n_points = [len(c) for c in clouds]; stride = (max(n_points) + 63) // 64 * 64
host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
for b, c in enumerate(clouds): host[b, :len(c)] = api.pack16(c)
points = torch.from_numpy(host.view(np.uint8).reshape(B, stride, 16)).cuda()
org = np.zeros((B, 3), np.float32); bz = np.full(B, -1.73)

def run(ns, cold):
    per = B // ns
    segs = [api.GroundSegmentation().init(120.0, 0.33, n_slots=per, max_points=stride) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    outs = [None] * ns
    def step():
        for k, st in enumerate(streams):
            lo, hi = k * per, (k + 1) * per
            with torch.cuda.stream(st):
                if cold: segs[k].reset_maps(0, per, odom_z=0.0, persistent_only=True, on_torch_stream=True)
                outs[k] = segs[k].filter_batch(points[lo:hi], n_points[lo:hi], org[lo:hi], bz[lo:hi], out=outs[k])
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15
    for s in segs: s.close()
    return dt

for cold in (True, False):
    for ns in (1, 2, 4):
        dt = run(ns, cold)
        print(f"B={B} contexts/streams={ns} {'cold' if cold else 'warm'}: {dt*1e3:.3f} ms/step  {B/dt:.0f} clouds/s", flush=True)
