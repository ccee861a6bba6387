"""Per-kernel ms of BASELINE configs[3] (2.1 M points, 1000 x 1000 grid @ 0.2 m) at 128 clouds per launch and for one cloud, for A/B
runs on the GPU box:  [GROUNDGRID_HIP_LIB=...] python tools/ab_config4.py [batch] [tag]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groundgrid_amd import api, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tag = sys.argv[2] if len(sys.argv) > 2 else ""
base = synth.os128_cloud_fast(seed=20240113)
n, stride = len(base), (len(base) + 63) // 64 * 64
seg = api.GroundSegmentation().init(200.0, 0.2, n_slots=B, max_points=stride)
seg.set_flags(profile=True)
if os.environ.get("SCAN_PARTS"):
    seg.debug_set_tuning("scan_parts", int(os.environ["SCAN_PARTS"]))
pts = torch.zeros((B, stride, 16), dtype=torch.uint8, device="cuda")
for b in range(B):
    ang = np.float32(2.0 * np.pi * b / B)
    c = synth.clone_cloud(base)
    c["x"] = (np.cos(ang) * base["x"] - np.sin(ang) * base["y"]).astype(np.float32)
    c["y"] = (np.sin(ang) * base["x"] + np.cos(ang) * base["y"]).astype(np.float32)
    pts[b, :n] = torch.from_numpy(api.pack16(c).view(np.uint8).reshape(-1, 16)).cuda()
res = {}
for nb in (B, 1):
    p = pts[:nb].contiguous()
    org, bz, ids = np.zeros((nb, 3), np.float32), np.full(nb, -1.73), np.arange(nb)
    out, shift = None, 0
    for k in range(2 + 4):
        if k == 2:
            seg.synchronize(); seg.kernel_times(reset=True)
        seg.reset_maps(0, nb, persistent_only=True, on_torch_stream=True)
        shift = (shift + bench.ROT) % nb
        out = seg.filter_batch(p, [n] * nb, org, bz, out=out, slots=((ids + shift) % nb).astype(np.int32))
    seg.synchronize()
    kt = {k: round(v[0] / max(1, v[1]), 4) for k, v in seg.kernel_times().items()}
    res[f"n1000_b{nb}"] = {"kernel_sum_ms": round(sum(kt.values()), 4), **kt}
print(json.dumps({"tag": tag, "lib": os.path.basename(os.environ.get("GROUNDGRID_HIP_LIB", "default")), **res}))
