"""Per-work-group start/end stamps of k_reduce (GG_K2_DEBUG=6): where do the CUs' work-group slots stand empty?
Needs a library built with the measurement switches: tools/build_variant.sh k2inst "-DGG_INSTRUMENT", GROUNDGRID_HIP_LIB=groundgrid_amd/variants/lib_k2inst.so
(the production k_reduce is compiled without them since round 5)."""
import os, sys, ctypes as C
os.environ["GG_K2_DEBUG"] = "6"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from groundgrid_amd import api, _lib
from groundgrid_amd.dist import common_stride

B = 1024
dev = torch.device("cuda", 0)
clouds = bench.make_clouds(B, 0)
npts = [len(c) for c in clouds]
stride = common_stride(max(npts), device=dev)
seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
seg.set_flags(profile=True)
host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
for b, c in enumerate(clouds):
    host[b, :len(c)] = api.pack16(c)
pts = torch.from_numpy(host.view(np.uint8).reshape(B, stride, 16)).cuda()
org = np.zeros((B, 3), np.float32); bz = np.full(B, -1.73)
L = _lib.load()
L.gg_debug_k2_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
out = None
for _ in range(4):
    seg.reset_maps(0, B, odom_z=0.0, persistent_only=True, on_torch_stream=True)
    out = seg.filter_batch(pts, npts, org, bz, out=out)
seg.synchronize()
kt = seg.kernel_times()
print("k_reduce ms", kt['k_reduce'][0] / max(1, kt['k_reduce'][1]))
n = 65536
buf = (C.c_ulonglong * (n * 4))()
assert L.gg_debug_k2_trace(seg._ctx, buf, n) == 0
v = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "k2_trace.npy"), v)
start, end, cu, kind = v[:, 0], v[:, 1], v[:, 2], v[:, 3] & 1
dur = end - start
print("work-groups", n, "dense", int(kind.sum()), "mean ticks (10 ns): dense %.0f light %.0f" % (dur[kind == 1].mean(), dur[kind == 0].mean()))
t0 = start.min(); span = end.max() - t0
print("kernel span %.3f ms" % (span * 1e-5))
nb = 30
edges = np.linspace(0, span, nb + 1)
for name, sel in (("all", np.ones(n, bool)), ("dense", kind == 1), ("light", kind == 0)):
    res = []
    for k in range(nb):
        lo, hi = edges[k] + t0, edges[k + 1] + t0
        res.append(np.clip(np.minimum(end[sel], hi) - np.maximum(start[sel], lo), 0, None).sum() / (hi - lo) / 256)
    print("residents per CU over the span (%s):" % name, " ".join("%.1f" % r for r in res))
xcc = cu >> 7
for x in range(8):
    s_ = xcc == x
    print("XCC", x, "first start %.3f last end %.3f ms, WG-time %.1f ms" % ((start[s_].min() - t0) * 1e-5, (end[s_].max() - t0) * 1e-5, dur[s_].sum() * 1e-5))
item = np.arange(n)
cloud_first = start.reshape(-1)[:]  # items are (cloud, group) in xcd-contiguous order
