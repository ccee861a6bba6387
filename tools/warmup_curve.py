"""Step time of the first frames after a map reset (cold -> warm), default batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from groundgrid_amd import api
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
clouds = bench.make_clouds(B, 0); n_points = [len(c) for c in clouds]; stride = (max(n_points) + 63) // 64 * 64
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
seg.set_flags(profile=True)
host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
for b, c in enumerate(clouds): host[b, :len(c)] = api.pack16(c)
points = torch.from_numpy(host.view(np.uint8).reshape(B, stride, 16)).cuda()
org = np.zeros((B, 3), np.float32); bz = np.full(B, -1.73); out = None
for rep in range(2):
    for b in range(B): seg.map(b).reset()
    seg.synchronize(); torch.cuda.synchronize()
    for f in range(6):
        seg.kernel_times(reset=True)
        t0 = time.perf_counter(); out = seg.filter_batch(points, n_points, org, bz, out=out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        kt = seg.kernel_times(reset=True)
        if rep: print(f"frame {f}: {dt*1e3:7.2f} ms  classify {kt['k_classify'][0]:.2f} reduce {kt['k_reduce'][0]:.2f} spiral {kt['k_sweep'][0]:.2f} label {kt['k_label'][0]:.2f}")
