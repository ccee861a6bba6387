"""Where k_reduce's time goes (GG_K2_DEBUG=9: cycle counters summed over the tiles of one launch), on the GPU box.
Needs a library built with the measurement switches: tools/build_variant.sh k2inst "-DGG_INSTRUMENT", GROUNDGRID_HIP_LIB=groundgrid_amd/variants/lib_k2inst.so
(the production k_reduce is compiled without them since round 5)."""
import os, sys, ctypes as C
os.environ["GG_K2_DEBUG"] = "9"
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
    out = None
    L = _lib.load()
    L.gg_debug_k2_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 64)()
    for _ in range(3):
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize(); seg.kernel_times(reset=True)
    assert L.gg_debug_k2_phases(seg._ctx, buf, 1) == 0
    out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize()
    assert L.gg_debug_k2_phases(seg._ctx, buf, 1) == 0
    kt = seg.kernel_times()
    v = np.array(list(buf), dtype=np.float64)
    ms = kt['k_reduce'][0] / max(1, kt['k_reduce'][1])
    print(f"batch {batch}: k_reduce {ms:.3f} ms")
    nd = max(v[0], 1)
    print(f"  dense tiles {int(v[0])} ({v[0]/batch:.1f}/cloud), records/tile {v[1]/nd:.0f}; cycles per tile: count {v[2]/nd:.0f} scan {v[3]/nd:.0f} place {v[4]/nd:.0f} chains {v[5]/nd:.0f} write {v[6]/nd:.0f}  total {v[2:7].sum()/nd:.0f}")
    print(f"  chain phase per wave (own time): " + " ".join(f"w{w}={v[16+w]/nd:.0f}" for w in range(4)) + f"   split tiles {int(v[20])}")
    nl = max(v[8], 1)
    print(f"  light tiles {int(v[8])} ({v[8]/batch:.1f}/cloud), records/tile {v[9]/nl:.0f}, cycles/tile {v[10]/nl:.0f};  reset-only tiles {int(v[12])}, cycles {v[13]/max(v[12],1):.0f}")
    print(f"  light tile split: head (lookups, records, count, place) {v[29]/nl:.0f}, chains + layer writes {v[28]/nl:.0f}")
    print(f"  work-groups: dense {int(v[24])} x {v[25]/max(v[24],1):.0f} cycles, light {int(v[26])} x {v[27]/max(v[26],1):.0f} cycles")
    seg.close()

for b in (1, 64, 1024):
    run(b)
