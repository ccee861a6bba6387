"""(GG_K3_DEBUG needs a library built with -DGG_INSTRUMENT: tools/build_variant.sh inst "-DGG_INSTRUMENT", GROUNDGRID_HIP_LIB=.../lib_inst.so)
k_patch per-launch time at a few batch sizes (HIP events inside the library), on the GPU box.  GG_K3_DEBUG=1/2/3 cuts the
kernel short (results are then wrong: timing only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth

def run(batch, same, steps=6):
    clouds = [synth.hdl64_cloud(seed=20240113 + (0 if same else k)) for k in range(min(batch, 4))]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=batch, max_points=stride)
    seg.set_flags(profile=True)
    host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
    n = []
    for b in range(batch):
        c = clouds[b % len(clouds)]
        host[b, :len(c)] = api.pack16(c); n.append(len(c))
    pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
    org = np.zeros((batch, 3), np.float32); bz = np.full(batch, -1.73)
    out = None
    for _ in range(3):
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize(); seg.kernel_times(reset=True)
    for _ in range(steps):
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize()
    kt = seg.kernel_times()
    seg.close()
    return {k: v[0] / max(1, v[1]) for k, v in kt.items()}

print("GG_K3_DEBUG =", os.environ.get("GG_K3_DEBUG"))
cases = ((1024, False),) if len(sys.argv) > 1 and sys.argv[1] == 'big' else ((1, True), (8, True), (8, False), (64, False), (1024, False))
for batch, same in cases:
    kt = run(batch, same)
    print(f"batch {batch:5d} same={same}: patch {kt['k_patch']:.4f} ms", flush=True)
