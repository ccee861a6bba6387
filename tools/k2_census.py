"""Which CUs ran k_reduce's work-groups, how many at a time and for how long (GG_K2_DEBUG=5), on the GPU box.
Needs a library built with the measurement switches: tools/build_variant.sh k2inst "-DGG_INSTRUMENT", GROUNDGRID_HIP_LIB=groundgrid_amd/variants/lib_k2inst.so
(the production k_reduce is compiled without them since round 5)."""
import os, sys, ctypes as C
os.environ["GG_K2_DEBUG"] = "5"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from groundgrid_amd import api, synth, _lib

def run(batch):
    clouds = [synth.hdl64_cloud(seed=20240113 + k) for k in range(min(batch, 4))]
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
    L = _lib.load()
    L.gg_debug_k2_census.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * (1024 * 8))()
    out = None
    for _ in range(3):
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize(); seg.kernel_times(reset=True)
    assert L.gg_debug_k2_census(seg._ctx, buf, 1) == 0
    out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize()
    assert L.gg_debug_k2_census(seg._ctx, buf, 1) == 0
    kt = seg.kernel_times()
    ms = kt['k_reduce'][0] / max(1, kt['k_reduce'][1])
    v = np.array(list(buf), dtype=np.uint64).reshape(1024, 8)
    used = v[:, 0] > 0
    u = v[used].astype(np.float64)
    span = u[:, 5] - u[:, 4]
    print(f"batch {batch}: k_reduce {ms:.3f} ms; CUs that ran work-groups: {used.sum()}")
    print(f"  work-groups per CU: min {u[:,0].min():.0f} mean {u[:,0].mean():.1f} max {u[:,0].max():.0f};  peak residents per CU: min {u[:,2].min():.0f} mean {u[:,2].mean():.2f} max {u[:,2].max():.0f}")
    print(f"  per CU: busy span (first start .. last end) mean {span.mean():.0f} max {span.max():.0f} cycles; work-group cycles / span = average residents: mean {(u[:,3]/span).mean():.2f} min {(u[:,3]/span).min():.2f} max {(u[:,3]/span).max():.2f}")
    xcc = (np.nonzero(used)[0] >> 7)
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print(f"  XCC {x}: {sel.sum()} CUs, work-groups {u[sel,0].sum():.0f}, span {span[sel].mean():.0f}, residents {(u[sel,3]/span[sel]).mean():.2f}")
    seg.close()

for b in (64, 1024):
    run(b)
