"""Run a few steps at one batch size (for rocprofv3): python tools/k4_run.py <batch> <levels 0|1> [steps]"""
import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth
batch, levels = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
clouds = [synth.hdl64_cloud(seed=20240113 + k) for k in range(min(batch, 4))]
stride = (max(len(c) for c in clouds) + 63) // 64 * 64
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=batch, max_points=stride)
host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE); n = []
for b in range(batch):
    c = clouds[b % len(clouds)]; host[b, :len(c)] = api.pack16(c); n.append(len(c))
pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
out = None
for _ in range(steps):
    out = seg.filter_batch(pts, n, np.zeros((batch, 3), np.float32), np.full(batch, -1.73), out=out)
seg.synchronize()
