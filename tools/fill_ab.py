"""Time of gg_reset_maps (persistent layers of 1024 maps) on the GPU box; GG_FILL_GROUP selects the fill mask's granularity."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from groundgrid_amd import api

B = 1024
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=1024)
for _ in range(3):
    seg.reset_maps(0, B, odom_z=0.0, persistent_only=True)
seg.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    seg.reset_maps(0, B, odom_z=0.0, persistent_only=True)
seg.synchronize()
print(f"GG_FILL_GROUP={os.environ.get('GG_FILL_GROUP', 'default')}: reset of {B} maps {(time.perf_counter() - t0) / K * 1e3:.3f} ms")
seg.close()
