"""gg_filter_cloud with k_label writing its results straight into pinned host memory (results_direct = 1, the default) against the copy behind the kernel (0):
   python tools/direct_probe.py -> JSON, ms per call (best of 3 passes), the arms alternate"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth
import tools.host_call_probe as h

clouds = [synth.hdl64_cloud(seed=20240113 + k) for k in range(8)]
stride = (max(len(c) for c in clouds) + 63) // 64 * 64
org = (0.0, 0.0, 0.0)
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=stride)
seq = [clouds[k % 8] for k in range(64)]
for c in seq[:8]:
    seg.filter_cloud(c, org, -1.73)
plain = seg.alloc_layers(register=False)
res = {}
for rep in range(2):
    for direct in (1, 0):
        seg.debug_set_tuning("results_direct", direct)
        res[f"sync_ms_direct{direct}_{rep}"] = h.best(lambda c: seg.filter_cloud(c, org, -1.73, reuse_buffers=True), seq)
        res[f"fused_all_layers_ms_direct{direct}_{rep}"] = h.best(lambda c: seg.filter_cloud_with_layers(c, org, -1.73, plain, reuse_buffers=True), seq[:32])
seg.debug_set_tuning("results_direct", 1)
for rep in range(2):
    for pieces in (2, 1, 3, 4):
        seg.debug_set_tuning("upload_pieces", pieces)
        res[f"sync_ms_pieces{pieces}_{rep}"] = h.best(lambda c: seg.filter_cloud(c, org, -1.73, reuse_buffers=True), seq)
seg.close()
print(json.dumps(res))
