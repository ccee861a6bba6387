"""Experiment: does splitting a batch over two HIP streams let the sweep (one 640-thread work-group per cloud, chain-bound)
overlap with the throughput-bound kernels of the other half?  usage: python tools/two_stream.py [B] [n_streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from groundgrid_amd import api
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
clouds = bench.make_clouds(B, 0)
n_points = [len(c) for c in clouds]; stride = max(n_points)
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
for b, c in enumerate(clouds): host[b, :len(c)] = api.pack16(c)
points = torch.from_numpy(host.view(np.uint8).reshape(B, stride, 16)).cuda()
org = np.zeros((B, 3), np.float32); bz = np.full(B, -1.73)
streams = [torch.cuda.Stream() for _ in range(NS)]
per = B // NS
outs = [None] * NS
def step():
    for k, st in enumerate(streams):
        lo, hi = k * per, (k + 1) * per
        with torch.cuda.stream(st):
            outs[k] = seg.filter_batch(points[lo:hi], n_points[lo:hi], org[lo:hi], bz[lo:hi], first_slot=lo, out=outs[k])
for _ in range(4): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(15): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15
print(f"B={B} streams={NS}: {dt*1e3:.3f} ms/step  {B/dt:.0f} clouds/s")
