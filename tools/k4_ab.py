"""Per-launch kernel times at several batch sizes (HIP events inside the library), on the GPU box."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth

def run(batch, levels, steps=8):
    clouds = [synth.hdl64_cloud(seed=20240113 + k) for k in range(min(batch, int(os.environ.get("NCLOUDS", "4"))))]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=batch, max_points=stride)
    seg.set_flags(profile=True, minimal_layers=bool(os.environ.get('MINIMAL')))
    host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
    n = []
    for b in range(batch):
        c = clouds[b % len(clouds)]
        host[b, :len(c)] = api.pack16(c); n.append(len(c))
    pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
    org = np.zeros((batch, 3), np.float32); bz = np.full(batch, -1.73)
    out = None
    for _ in range(3):
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize(); seg.kernel_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kt = seg.kernel_times()
    seg.close()
    return dt * 1e3, {k: v[0] / max(1, v[1]) for k, v in kt.items()}

for batch in (1, 8, 64, 256, 1024):
    ms, kt = run(batch, False)
    print(f"batch {batch:5d}: step {ms:8.3f} ms  k_sweep {kt['k_sweep']:8.4f} ms   " + " ".join(f"{k[2:]}={v:.3f}" for k, v in kt.items() if k != 'k_sweep'), flush=True)
