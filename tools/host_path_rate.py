"""PCIe-inclusive rate of the drop-in entry point gg_filter_cloud (host buffers in / out, synchronous, one cloud at a time) and
of the two-deep gg_filter_cloud_async pipeline; GG_HOST_TIMING=1 prints where the host call spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from groundgrid_amd import api, synth
c = synth.hdl64_cloud()
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(c))
for _ in range(5): seg.filter_cloud(c, (0, 0, 0), -1.73)
N = 300
t = time.perf_counter()
for _ in range(N): out = seg.filter_cloud(c, (0, 0, 0), -1.73)
dt = (time.perf_counter() - t) / N
print(f"gg_filter_cloud (host in/out, {len(c)} pts): {dt*1e3:.3f} ms/cloud = {1/dt:.1f} clouds/s (includes packing, H2D 16 B/pt, D2H 5 B/pt, host assembly of the returned cloud)")
t = time.perf_counter()
tick = seg.filter_cloud_async(c, (0, 0, 0), -1.73)
for k in range(N):
    nxt = seg.filter_cloud_async(c, (0, 0, 0), -1.73) if k + 1 < N else None
    seg.filter_cloud_wait(tick)
    tick = nxt
dt = (time.perf_counter() - t) / N
print(f"two clouds deep: {dt*1e3:.3f} ms/cloud = {1/dt:.1f} clouds/s")
