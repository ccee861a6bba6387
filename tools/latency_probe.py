"""Single-launch kernel times (HIP events inside the library) for small batches: the latency side of the path.
   python tools/latency_probe.py            -> JSON: per geometry / batch the per-kernel ms and the step ms"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth


def run(length, res, cloud, batch, steps=12, cold=True):
    stride = (len(cloud) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, res, n_slots=batch, max_points=stride)
    seg.set_flags(profile=True)
    host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
    for b in range(batch):
        host[b, : len(cloud)] = api.pack16(cloud)
    pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
    n, org, bz = [len(cloud)] * batch, np.zeros((batch, 3), np.float32), np.full(batch, -1.73)
    out = None
    for k in range(3 + steps):
        if k == 3:
            seg.synchronize(); seg.kernel_times(reset=True); t0 = time.perf_counter()
        if cold:
            seg.reset_maps(0, batch, persistent_only=True, on_torch_stream=True)
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kt = {k: round(v[0] / max(1, v[1]), 4) for k, v in seg.kernel_times().items()}
    seg.close()
    return {"ms_per_step": round(dt * 1e3, 4), **kt}


if __name__ == "__main__":
    res = {"lib": os.environ.get("GROUNDGRID_HIP_LIB", "default")}
    c2 = synth.hdl64_cloud(seed=20240113)
    if not os.environ.get("SKIP_SMALL"):
        for b in [int(v) for v in os.environ.get("BATCHES_SMALL", "1,8,64").split(",")]:
            res[f"n364_b{b}"] = run(120.0, 0.33, c2, b)
    if not os.environ.get("SKIP_BIG"):
        c4 = synth.os128_cloud_fast(seed=20240113)
        for b in [int(v) for v in os.environ.get("BATCHES_BIG", "1,8").split(",")]:
            res[f"n1000_b{b}"] = run(200.0, 0.2, c4, b, steps=6)
    print(json.dumps(res))
