"""GG_FLAG_CONCURRENT_HALVES on the headline workload: ms per 1024-cloud step, cold and warm, flag off / on (/ on without the fork: unsafe, diagnostic).
   python tools/halves_probe.py [batch] [steps]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groundgrid_amd import api

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
clouds = bench.make_clouds(batch, 0, n_scenes=32)
n = [len(c) for c in clouds]
stride = (max(n) + 63) // 64 * 64
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=batch, max_points=stride)
host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
for b, c in enumerate(clouds):
    host[b, : len(c)] = api.pack16(c)
pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
org, bz = np.zeros((batch, 3), np.float32), np.full(batch, -1.73)
ids = np.arange(batch)
res = {}
side = torch.cuda.Stream() if os.environ.get("STREAM", "side") == "side" else torch.cuda.default_stream()
for name, halves, nofork in (("off", False, 0), ("on", True, 0), ("off2", False, 0), ("on2", True, 0)):
    seg.set_flags(concurrent_halves=halves)
    seg.debug_set_tuning("halves_no_fork", nofork)
    for mode in ("cold", "warm"):
        out, shift = None, 0
        with torch.cuda.stream(side):
            for k in range(3 + steps):
                if k == 3:
                    seg.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
                if mode == "cold":
                    seg.reset_maps(0, batch, persistent_only=True, on_torch_stream=True)
                    shift = (shift + bench.ROT) % batch
                out = seg.filter_batch(pts, n, org, bz, out=out, slots=((ids + shift) % batch).astype(np.int32))
            seg.synchronize(); torch.cuda.synchronize()
        res[f"{name}_{mode}"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
print(json.dumps(res))
