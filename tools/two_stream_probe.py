"""Does the batch run faster as two half batches on two streams (two contexts) than as one launch sequence?
python tools/two_stream_probe.py [batch] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from groundgrid_amd import api
from groundgrid_amd.dist import common_stride

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
clouds = bench.make_clouds(B, 0)
npts = [len(c) for c in clouds]
stride = common_stride(max(npts), device=dev)

def to_device(cl):
    host = np.zeros((len(cl), stride), dtype=api.POINT16_DTYPE)
    for b, c in enumerate(cl):
        host[b, : len(c)] = api.pack16(c)
    return torch.from_numpy(host.view(np.uint8).reshape(len(cl), stride, 16)).to(dev).contiguous()

def run(parts, stagger_cycles=0, halves=False, default_stream=False, rotate=False):
    per = B // parts
    segs, pts, outs, streams = [], [], [], []
    for p in range(parts):
        s = api.GroundSegmentation().init(120.0, 0.33, n_slots=per, max_points=stride, device=0)
        s.set_flags(profile=False)  # (halves: tools/experiments/concurrent_halves_in_library.patch adds that flag)
        segs.append(s)
        pts.append(to_device(clouds[p * per:(p + 1) * per]))
        outs.append(None)
        streams.append(torch.cuda.default_stream(dev) if default_stream else torch.cuda.Stream(device=dev))
    org = np.zeros((per, 3), dtype=np.float32)
    bz = np.full(per, -1.73)
    state = {"shift": 0}
    ids = np.arange(per)
    def step():
        if rotate:
            state["shift"] = (state["shift"] + bench.ROT) % per
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                segs[p].reset_maps(0, per, odom_z=0.0, persistent_only=True, on_torch_stream=True)
                outs[p] = segs[p].filter_batch(pts[p], npts[p * per:(p + 1) * per], org, bz, out=outs[p],
                                               slots=((ids + state["shift"]) % per).astype(np.int32) if rotate else None)
    if stagger_cycles:
        for p in range(1, parts):
            with torch.cuda.stream(streams[p]):
                torch.cuda._sleep(int(stagger_cycles * p / parts))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    if stagger_cycles:
        for p in range(1, parts):
            with torch.cuda.stream(streams[p]):
                torch.cuda._sleep(int(stagger_cycles * p / parts))
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    for s in segs:
        s.close()
    return dt

for parts in (1, 2, 1, 2, 4):
    dt = run(parts)
    print(f"parts={parts}: {dt*1e3:.3f} ms per {B} clouds = {B/dt:.0f} clouds/s", flush=True)
