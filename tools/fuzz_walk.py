"""Fuzz of k_classify's line-of-sight walk (src/GroundSegmentation.cpp:243-275) against the oracle: maps that meet unrelated scenes, points
pushed under the stored terrain in every density (none / a few / most lanes of a window: the dealt-items path and the ray-per-lane path),
sensor origins off the map centre and at heights that put rays on both sides of vec.z < -0.01, rays that leave the map, map positions off
the origin.  python tools/fuzz_walk.py [first] [last]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from groundgrid_amd import api, synth
from oracle import oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
n_outliers = n_candidates_frames = 0
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    length, res = [(120.0, 0.33), (60.0, 0.33), (80.0, 0.25), (40.0, 0.2)][int(rng.integers(0, 4))]
    k = np.float32(length / 120.0)
    scenes = []
    for s in range(3):
        c = synth.clone_cloud(synth.hdl64_cloud(seed=7000 + 13 * seed + s, n_az=int(rng.integers(150, 420)), order=str(rng.choice(["ring", "azimuth"]))))
        c["x"] *= k
        c["y"] *= k
        frac = float(rng.choice([0.0, 0.02, 0.2, 0.7, 1.0]))
        sel = rng.random(len(c)) < frac
        c["z"][sel] -= rng.uniform(0.21, 3.0, int(sel.sum())).astype(np.float32)
        scenes.append(c)
    pos = tuple(np.round(rng.uniform(-2, 2, size=2), 2))
    cap = max(len(c) for c in scenes)
    seg = api.GroundSegmentation().init(length, res, n_slots=1, max_points=cap)
    ref = oracle.OracleMap(length, res, pos=pos)
    seg.map(0).reset(pos=pos)
    try:
        for frame in range(7):
            c = scenes[int(rng.integers(0, 3))]
            origin = (float(pos[0] + rng.uniform(-0.3, 0.3) * length), float(pos[1] + rng.uniform(-0.3, 0.3) * length), float(rng.choice([0.0, 0.3, -1.5, -1.72, -2.5, 4.0])))
            shifted = synth.clone_cloud(c)
            shifted["x"] += np.float32(pos[0])
            shifted["y"] += np.float32(pos[1])
            _, labels, index = seg.filter_cloud(shifted, origin, -1.73, return_details=True)
            r = ref.filter_cloud(shifted, origin, -1.73)
            cls, cell = seg.point_classes(len(shifted))
            n_outliers += int((r["cls"] == oracle.OUTLIER).sum())
            n_candidates_frames += 1 if (r["cls"] == oracle.OUTLIER).any() else 0
            assert np.array_equal(cls, r["cls"]), (frame, "classes", int((cls != r["cls"]).sum()))
            assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), (frame, "labels")
            for name in ("ground", "groundpatch", "points", "pointsRaw", "variance"):
                assert np.array_equal(seg.map(0)[name], ref.layer(name), equal_nan=True), (frame, name)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:200], flush=True)
    seg.close()
print("walk fuzz done, seeds", first, "..", last, "failures:", bad, "-- outliers found by the walks:", n_outliers, "in", n_candidates_frames, "of", 7 * (last - first), "frames")
