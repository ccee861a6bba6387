"""Per-kernel event timings (GG_FLAG_PROFILE) for a few cloud types and batch sizes (debug aid, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from groundgrid_amd import api, synth

def run(name, clouds, steps=10, warm=3, minimal=False, length=120.0, res=0.33, dbg=0):
    B = len(clouds); stride = max(len(c) for c in clouds)
    seg = api.GroundSegmentation().init(length, res, n_slots=B, max_points=stride)
    seg.set_flags(minimal_layers=minimal, profile=True)
    if dbg:
        from groundgrid_amd import _lib
        seg._L.gg_set_flags(seg._ctx, _lib.GG_FLAG_PROFILE | dbg)
    host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
    for b, c in enumerate(clouds): host[b, :len(c)] = api.pack16(c)
    pts = torch.from_numpy(host.view(np.uint8).reshape(B, stride, 16)).cuda()
    org = np.zeros((B,3),np.float32); bz = np.full(B,-1.73)
    out=None
    for _ in range(warm): out = seg.filter_batch(pts, [len(c) for c in clouds], org, bz, out=out)
    torch.cuda.synchronize(); seg.kernel_times(reset=True)
    for _ in range(steps): out = seg.filter_batch(pts, [len(c) for c in clouds], org, bz, out=out)
    torch.cuda.synchronize()
    kt = seg.kernel_times(reset=True)
    print(f"{name:28s} B={B:3d} n={stride:8d} " + " ".join(f"{k[2:]}={v[0]/max(1,v[1])*1e3:8.1f}us" for k,v in kt.items()), flush=True)
    seg.close()

if __name__ == "__main__":
    which = sys.argv[1:] or ["hdl", "rand", "hdlaz", "min"]
    hdl = synth.hdl64_cloud()
    if "hdl" in which: run("hdl64 ring-major", [hdl])
    if "min" in which: run("hdl64 minimal layers", [hdl], minimal=True)
    if "hdlaz" in which: run("hdl64 azimuth-major", [synth.hdl64_cloud(order="azimuth")])
    if "rand" in which: run("uniform random 120k", [synth.random_cloud(125000, seed=1, extent=59.0)])
    if "b8" in which: run("hdl64 x8", [synth.hdl64_cloud(seed=s) for s in range(8)])
    if "dbg" in which:
        cl = [synth.hdl64_cloud(seed=s) for s in range(8)] * 8
        for d, nm in ((0, "base"), (0x100, "no LDS atomics"), (0x200, "no ground gather"), (0x400, "no rec store"), (0x800, "no hist flush"), (0xF00, "none of them")):
            run("B64 " + nm, cl, dbg=d)
    if "os" in which: run("os128 2.1M 1000^2", [synth.os128_cloud()], steps=3, warm=1, length=200.0, res=0.2)
    if "small" in which:
        run("hdl64 on 60 m map (182^2)", [hdl], length=60.0)
        run("hdl64 on 30 m map (90^2)", [hdl], length=30.0)
        run("hdl64 on 240 m map (728^2)", [hdl], length=240.0)
