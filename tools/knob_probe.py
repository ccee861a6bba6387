"""One cloud per launch / per host call under the values of one gg_debug_set_tuning key (ms, best of 3 passes, the arms alternate twice):
   python tools/knob_probe.py front 1 2 3        # the front end as three launches (the default), scan inside k_classify, one launch
   python tools/knob_probe.py scan_parts 0 2 4 8 # work-groups of k_scan per cloud"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth
import tools.host_call_probe as h
import time

KEY = sys.argv[1] if len(sys.argv) > 1 else "front"
VALUES = [int(v) for v in sys.argv[2:]] or [1, 2, 3]
clouds = [synth.hdl64_cloud(seed=20240113 + k) for k in range(4)]
stride = (max(len(c) for c in clouds) + 63) // 64 * 64
org = (0.0, 0.0, 0.0)
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=stride)
seq = [clouds[k % 4] for k in range(48)]
for c in seq[:6]:
    seg.filter_cloud(c, org, -1.73)
host = np.zeros((1, stride), dtype=api.POINT16_DTYPE)
host[0, : len(clouds[0])] = api.pack16(clouds[0])
pts = torch.from_numpy(host.view(np.uint8).reshape(1, stride, 16)).cuda()
n, o3, bz = [len(clouds[0])], np.zeros((1, 3), np.float32), np.full(1, -1.73)
side = torch.cuda.Stream()
res = {}
for rep in range(2):
    for shape in VALUES:
        seg.debug_set_tuning(KEY, shape)
        res[f"sync_ms_{KEY}{shape}_{rep}"] = h.best(lambda c: seg.filter_cloud(c, org, -1.73, reuse_buffers=True), seq)
        with torch.cuda.stream(side):
            out = None
            for _ in range(6):
                out = seg.filter_batch(pts, n, o3, bz, out=out)
            seg.synchronize()
            t = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(40):
                    out = seg.filter_batch(pts, n, o3, bz, out=out)
                seg.synchronize()
                t = min(t, (time.perf_counter() - t0) / 40)
        res[f"one_cloud_ms_{KEY}{shape}_{rep}"] = round(t * 1e3, 4)
seg.close()
print(json.dumps(res))
