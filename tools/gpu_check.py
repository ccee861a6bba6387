"""Stage-by-stage comparison of the HIP path with the oracle (debug aid, run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from groundgrid_amd import api, synth
from oracle import oracle

def cmp_layers(seg, ref, names=None, tag=""):
    ok = True
    for n in (names or oracle.LAYERS):
        a = seg.map(0)[n]; b = ref.layer(n)
        same = np.array_equal(a, b, equal_nan=True)
        if not same:
            d = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            print(f"  {tag} layer {n}: {len(d)} cells differ, first {d[:3].tolist()} gpu={[float(a[tuple(i)]) for i in d[:3]]} ref={[float(b[tuple(i)]) for i in d[:3]]}")
            ok = False
    return ok

def run(cloud, frames=3, origin=(0,0,0), base_z=-1.73, name=""):
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=max(len(cloud),1))
    ref = oracle.OracleMap(120.0, 0.33)
    allok = True
    for f in range(frames):
        t=time.time()
        out, labels, index = seg.filter_cloud(cloud, origin, base_z, return_details=True)
        dt=time.time()-t
        r = ref.filter_cloud(cloud, origin, base_z)
        cls, cell = seg.point_classes(len(cloud))
        ok_cls = np.array_equal(cls, r['cls']); ok_cell = np.array_equal(cell, r['cell'])
        ok_lab = np.array_equal(labels, r['label']); ok_idx = np.array_equal(index, r['index'])
        ok_out = out.tobytes() == r['out_points'].tobytes()
        print(f"{name} frame {f}: n={len(cloud)} t={dt*1e3:.2f}ms cls={ok_cls} cell={ok_cell} label={ok_lab} index={ok_idx} out={ok_out} "
              f"(labels diff {int((labels!=r['label']).sum())})")
        if not ok_out:
            a=out; b=r['out_points']
            print("   out len", len(a), len(b), "n_outl", int((r['cls']==2).sum()))
            if len(a)==len(b):
                ra=a.view(np.uint8).reshape(-1,32); rb=b.view(np.uint8).reshape(-1,32)
                bad=np.argwhere((ra!=rb).any(axis=1)).ravel()
                print("   bad rows", len(bad), bad[:5], ra[bad[:2]], rb[bad[:2]])
        ok_layers = cmp_layers(seg, ref, tag=f"f{f}")
        allok &= ok_cls and ok_cell and ok_lab and ok_idx and ok_out and ok_layers
    return allok

if __name__ == "__main__":
    ok = True
    ok &= run(synth.hdl64_cloud(seed=7, n_az=260), name="small")
    ok &= run(synth.random_cloud(20000, seed=3), name="random")
    ok &= run(synth.hdl64_cloud(), name="hdl64")
    ok &= run(synth.hdl64_cloud(order="azimuth"), name="hdl64-az")
    print("ALL OK" if ok else "MISMATCH")
