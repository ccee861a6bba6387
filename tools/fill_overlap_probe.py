"""Can the re-initialisation of a step's maps (memory-bound, 20 registers) hide behind the kernels of a step (issue-bound)?
Two sets of map slots take turns: while set A is filtered on the compute stream, set B is re-initialised on a side stream.
   python tools/fill_overlap_probe.py [batch]      (timing only: run on the GPU box)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groundgrid_amd import api

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
clouds = bench.make_clouds(batch, 0, n_scenes=8)
n = [len(c) for c in clouds]
stride = (max(n) + 63) // 64 * 64
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=2 * batch, max_points=stride)
host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
for b, c in enumerate(clouds):
    host[b, : len(c)] = api.pack16(c)
pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
org, bz = np.zeros((batch, 3), np.float32), np.full(batch, -1.73)
ids = np.arange(batch)
side = torch.cuda.Stream()
res = {}
for mode in ("serial", "overlapped"):
    out = [None, None]
    seg.reset_maps(0, 2 * batch, persistent_only=True, on_torch_stream=True)
    torch.cuda.synchronize()
    steps = 10
    for k in range(3 + steps):
        if k == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        half = k % 2
        slots = (half * batch + ids).astype(np.int32)
        if mode == "serial":
            seg.reset_maps(half * batch, batch, persistent_only=True, on_torch_stream=True)
            out[half] = seg.filter_batch(pts, n, org, bz, out=out[half], slots=slots)
        else:
            seg.debug_set_tuning("probe_unordered_streams", 1)  # (this script orders the two streams itself, with the events below)
            # the OTHER half's maps are re-initialised meanwhile (they were filtered in the step before: order the fill behind it)
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                seg.reset_maps((1 - half) * batch, batch, persistent_only=True, on_torch_stream=True)
                done = torch.cuda.Event(); done.record()
            out[half] = seg.filter_batch(pts, n, org, bz, out=out[half], slots=slots)
            torch.cuda.current_stream().wait_event(done)
    torch.cuda.synchronize(); seg.synchronize()
    res[mode] = round((time.perf_counter() - t0) / steps * 1e3, 3)
print(json.dumps({"batch": batch, "ms_per_step": res}))
