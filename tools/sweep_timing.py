"""Cycle counters of the sweep's wavefronts for one cloud (GG_SWEEP_TIMING=1): where does the single-cloud latency go?"""
import os, sys, ctypes as C
os.environ["GG_SWEEP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from groundgrid_amd import api, synth, _lib
cloud = synth.hdl64_cloud(seed=20240113)
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
for _ in range(3):
    seg.filter_cloud(cloud, (0, 0, 0), -1.73)
L = _lib.load()
out = (C.c_ulonglong * 64)()
L.gg_debug_sweep_timing.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert L.gg_debug_sweep_timing(seg._ctx, out) == 0
v = np.array(list(out), dtype=np.int64)[:48].reshape(12, 4)  # (the first 12 wavefronts; the corner phase marks follow)
W = 3 if os.environ.get("GG_SWEEP_WAVES") is None else int(os.environ["GG_SWEEP_WAVES"])
names = [f"{s}{w}" for w in range(W) for s in "ABCD"] + ["cornerAB", "cornerCD"]  # wavefront id = 4 w + side
names = names[:12]
t0 = v[:len(names), 0].min()
for k, nm in enumerate(names):
    print(f"{nm:9s} start {v[k,0]-t0:9d} end {v[k,1]-t0:9d} cycles  polling {v[k,2]:9d}  waits {v[k,3]:6d}")
m = np.array(list(out), dtype=np.int64)[48:60].reshape(2, 6)
print("corner phase cycles per ring (wave parity 0 / 1): load-consume+decays, gets+X0, X1, Y0, publish", (m[:, 1:] / 180.0).round(0).tolist())
