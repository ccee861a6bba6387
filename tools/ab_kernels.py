"""Per-kernel ms of the headline workload (cold maps, clouds rotating over the slots) for A/B runs on the GPU box:
   [GROUNDGRID_HIP_LIB=...] [GG_K2_PER_CLOUD=..] python tools/ab_kernels.py [batch] [steps] [tag]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groundgrid_amd import api

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tag = sys.argv[3] if len(sys.argv) > 3 else ""
clouds = bench.make_clouds(batch, 0, n_scenes=int(os.environ.get("N_SCENES", "8")))
n = [len(c) for c in clouds]
stride = (max(n) + 63) // 64 * 64
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=batch, max_points=stride)
seg.set_flags(profile=True)
if os.environ.get("SCAN_PARTS"):
    seg.debug_set_tuning("scan_parts", int(os.environ["SCAN_PARTS"]))
host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
for b, c in enumerate(clouds):
    host[b, : len(c)] = api.pack16(c)
pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
org, bz = np.zeros((batch, 3), np.float32), np.full(batch, -1.73)
ids = np.arange(batch)
out, shift = None, 0
res = {}
for mode in os.environ.get("MODES", "cold,warm").split(","):  # mixed = warm maps, the clouds rotate: unrelated scenes meet (the ray-walk stress)
    for k in range(3 + steps):
        if k == 3:
            seg.synchronize(); seg.kernel_times(reset=True); t0 = time.perf_counter()
        if mode == "cold":
            seg.reset_maps(0, batch, persistent_only=True, on_torch_stream=True)
        if mode in ("cold", "mixed"):
            shift = (shift + bench.ROT) % batch
        out = seg.filter_batch(pts, n, org, bz, out=out, slots=((ids + shift) % batch).astype(np.int32))
    seg.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res[mode] = {"ms_per_step": round(dt * 1e3, 3), **{k: round(v[0] / max(1, v[1]), 3) for k, v in seg.kernel_times().items()}}
print(json.dumps({"tag": tag, "lib": os.path.basename(os.environ.get("GROUNDGRID_HIP_LIB", "default")), "batch": batch, **res}))
