"""k_sweep per launch for small batches: the pair sweep (sweep_pair.h) against k_sweep, on the GPU box.
   python tools/pair_ab.py [batches, default 1,2,8,16]  (GEOM=big adds the 1000 x 1000 map)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth
from oracle import oracle


def run(length, res, cloud, batch, knobs, steps=20, check=False):
    stride = (len(cloud) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, res, n_slots=batch, max_points=stride)
    seg.set_flags(profile=True)
    for k, v in knobs.items():
        seg.debug_set_tuning(k, v)
    host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
    for b in range(batch):
        host[b, : len(cloud)] = api.pack16(cloud)
    pts = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
    n, org, bz = [len(cloud)] * batch, np.zeros((batch, 3), np.float32), np.full(batch, -1.73)
    out = None
    for k in range(4 + steps):
        if k == 4:
            seg.synchronize(); seg.kernel_times(reset=True); t0 = time.perf_counter()
        out = seg.filter_batch(pts, n, org, bz, out=out)
    seg.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kt = {k: round(v[0] / max(1, v[1]), 4) for k, v in seg.kernel_times().items()}
    ok = None
    if check:
        ref = oracle.OracleMap(length, res)
        for _ in range(4 + steps):
            ref.filter_cloud(cloud, (0.0, 0.0, 0.0), -1.73)
        g = seg.map(batch - 1)["ground"]
        w = seg.map(batch - 1)["groundpatch"]
        ok = bool(np.array_equal(g, ref.layer("ground"), equal_nan=True) and np.array_equal(w, ref.layer("groundpatch"), equal_nan=True))
        if not ok and os.environ.get("PAIR_AB_VERBOSE"):
            n = g.shape[0]; c = n // 2 - 1
            bad = np.argwhere(~((g == ref.layer("ground")) | (np.isnan(g) & np.isnan(ref.layer("ground")))))
            from collections import Counter
            cnt = Counter(); first = {}
            for x, y in bad.tolist():
                dx, dy = x - c, y - c; r = max(abs(dx), abs(dy))
                if dx == -r and dy < r: side, k = 'A', y - (c - r)
                elif dx == r: side, k = 'C', (c + r) - y
                elif dy == -r: side, k = 'B', x - (c - r)
                else: side, k = 'D', (c + r) - x
                cnt[(r, side)] += 1; first[(r, side)] = min(first.get((r, side), 10**9), k)
            print("   ground differs in", len(bad), "cells; w differs in", int((w != ref.layer("groundpatch")).sum()), "; by (ring, side): count, first position:",
                  [(k, cnt[k], first[k]) for k in sorted(cnt)][:24], flush=True)
    seg.close()
    return {"ms_per_step": round(dt * 1e3, 4), "k_sweep": kt.get("k_sweep"), "parity": ok}


if __name__ == "__main__":
    batches = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,8,16").split(",")]
    c2 = synth.hdl64_cloud(seed=20240113)
    arms = {"k_sweep": {"sweep_pair": 2}, "pair_2wg": {"sweep_pair": 0, "sweep_pair_wgs": 2}, "pair_1wg": {"sweep_pair": 0, "sweep_pair_wgs": 1}}
    for b in batches:
        for name, knobs in arms.items():
            r = run(120.0, 0.33, c2, b, knobs, check=(b == batches[0]))
            print(f"n364 batch {b:3d} {name:9s} k_sweep {r['k_sweep']:.4f} ms  step {r['ms_per_step']:.4f} ms  parity {r['parity']}", flush=True)
    if os.environ.get("GEOM") == "big":
        c4 = synth.os128_cloud_fast(seed=20240113)
        for b in (1, 4):
            for name, knobs in arms.items():
                if name == "pair_1wg":
                    continue
                r = run(200.0, 0.2, c4, b, knobs, steps=8, check=(b == 1))
                print(f"n1000 batch {b:3d} {name:9s} k_sweep {r['k_sweep']:.4f} ms  step {r['ms_per_step']:.4f} ms  parity {r['parity']}", flush=True)
