"""Batches of FRESH maps (gg_reset_maps leaves the (ground, confidence) layer unwritten; k_patch marks what it writes, k_sweep<FRESH> reads
nothing else) on random geometries -- odd and even sizes, one to four ring groups, one or several work-groups per cloud -- with random
clouds, heights and batch sizes, then a second call on the maps the first left and a third after a re-initialisation of a random subset
(a launch that mixes fresh and warm maps fills the fresh ones first): labels and the two persistent layers of EVERY map against the
oracle.  On the GPU box:  python tools/fuzz_fresh.py [first] [last]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth
from oracle import oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(9000 + seed)
    length, resolution = [(20.0, 0.2), (22.0, 0.33), (23.0, 0.33), (40.0, 0.33), (43.0, 0.33), (61.0, 0.25), (64.0, 0.33), (90.0, 0.33), (120.0, 0.33)][int(rng.integers(0, 9))]
    batch = int(rng.choice([130, 200, 257, 300]))
    z0 = float(np.float32(rng.uniform(-0.5, 0.5)))
    base = []
    for k in range(4):  # four scenes, rotated over the batch
        parts = []
        for j in range(int(rng.integers(3, 7))):
            centre = rng.uniform(-0.5 * length, 0.5 * length, size=2)
            spread = float(rng.choice([0.3, 1.5, 6.0, 20.0]))
            m = int(rng.integers(50, 4000))
            parts.append(np.column_stack([centre + rng.normal(0, spread, size=(m, 2)), rng.normal(rng.uniform(-2.5, 0.5), rng.choice([0.0, 0.02, 0.4]), size=m)]))
        pts = np.concatenate(parts).astype(np.float32)
        base.append(synth.make_cloud(pts, ring=rng.integers(0, 64, len(pts))))
    clouds = [base[b % 4] for b in range(batch)]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    try:
        seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
        halves, eager = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        seg.set_flags(concurrent_halves=halves, eager_layers=eager)
        knobs = {"sweep_waves": int(rng.choice([0, 1, 2, 3])), "front": int(rng.choice([0, 1, 2, 3])), "scan_parts": int(rng.choice([0, 2, 5])),
                 "halves_min_clouds": int(rng.choice([0, 64, 200]))}
        for key, value in knobs.items():
            if value:
                seg.debug_set_tuning(key, value)
        host = np.zeros((batch, stride), dtype=api.POINT16_DTYPE)
        for b, c in enumerate(clouds):
            host[b, : len(c)] = api.pack16(c)
        pts_d = torch.from_numpy(host.view(np.uint8).reshape(batch, stride, 16)).cuda()
        n = [len(c) for c in clouds]
        org = np.zeros((batch, 3), np.float32)
        seg.reset_maps(odom_z=z0)
        refs = [oracle.OracleMap(length, resolution, odom_z=z0) for _ in range(4)]  # (maps b, b + 4, ... see the same history)
        again = rng.random(batch) < 0.5
        again[:4] = [True, False, True, False]  # per scene: one history with, one without the re-initialisation
        refs2 = [oracle.OracleMap(length, resolution, odom_z=z0) for _ in range(4)]
        out = None
        for call in range(3):
            bz = float(rng.uniform(-2.0, -1.4))
            if call == 2:
                z1 = float(np.float32(rng.uniform(-0.5, 0.5)))
                for b in np.nonzero(again)[0]:
                    seg.reset_maps(first_slot=int(b), n_slots=1, odom_z=z1, persistent_only=True, on_torch_stream=True)
                for r in refs2:
                    r.set_layer("ground", np.full((r.rows, r.cols), np.float32(z1)))
                    r.set_layer("groundpatch", np.full((r.rows, r.cols), np.float32(0.0000001)))
            out = seg.filter_batch(pts_d, n, org, np.full(batch, bz), out=out)
            seg.batch_fence()
            torch.cuda.synchronize()
            labels = out.labels.cpu().numpy()
            res = [r.filter_cloud(base[k], (0.0, 0.0, 0.0), bz) for k, r in enumerate(refs)]
            res2 = [r.filter_cloud(base[k], (0.0, 0.0, 0.0), bz) for k, r in enumerate(refs2)]
            for b in range(batch):
                r, rr = (refs2[b % 4], res2[b % 4]) if again[b] else (refs[b % 4], res[b % 4])
                assert np.array_equal(labels[b, : n[b]], rr["label"]), f"call {call} cloud {b}: labels"
                for name in ("ground", "groundpatch"):
                    a, e = seg.map(b)[name], r.layer(name)
                    assert np.array_equal(a, e, equal_nan=True), f"call {call} cloud {b} layer {name}: {int((a != e).sum())} cells, first {np.argwhere(a != e)[:3].tolist()}"
        seg.close()
        if seed % 20 == 0:
            print("seed", seed, (length, resolution, batch, seg.rows), "ok", flush=True)
    except Exception as e:
        bad += 1
        print("seed", seed, (length, resolution, batch), "FAILED:", str(e)[:300], flush=True)
        try:
            print("   knobs", halves, eager, knobs, flush=True)
        except NameError:
            pass
print("done, failures:", bad)
