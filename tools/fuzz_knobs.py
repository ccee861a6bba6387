"""Random scenes (tests/test_gpu_parity.py::test_random_scenes_fuzz's generator) under random COMBINATIONS of the library's launch
shapes and flags -- points per wavefront chunk, front end in one / two / three launches, tile scan in parts, sweep parts, lazily
materialised layers -- three frames each, everything against the oracle.  On the GPU box:  python tools/fuzz_knobs.py [first] [last]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as t
from groundgrid_amd import api, synth
from oracle import oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    knobs = {"pw": int(rng.choice([0, 64, 256, 1024, 2048])), "front": int(rng.choice([1, 2, 3])), "scan_parts": int(rng.choice([0, 2, 5, 16])),
             "sweep_gpw": int(rng.choice([0, 1, 2])), "minimal": bool(rng.integers(0, 2)), "k2_per_cloud": int(rng.choice([0, 16, 64])),
             # round 6: which sweep takes the (single-cloud) launch -- the pair sweep (0; with sweep_gpw set: k_sweep), k_sweep (2), the throughput
             # pair sweep (4) -- and the pair sweeps' shapes
             "sweep_pair": int(rng.choice([0, 0, 2, 4])), "sweep_pair_wgs": int(rng.choice([0, 1])), "sweep_pair_waves": int(rng.choice([0, 1, 2, 3]))}
    length, resolution = [(20.0, 0.2), (40.0, 0.33), (64.0, 0.33), (120.0, 0.33), (150.0, 0.25), (200.0, 0.2)][int(rng.integers(0, 6))]
    parts = []
    for k in range(int(rng.integers(3, 8))):
        centre = rng.uniform(-0.55 * length, 0.55 * length, size=2)
        spread = float(rng.choice([0.05, 0.3, 1.5, 6.0, 20.0]))
        m = int(rng.integers(50, 30000 if k == 0 else 5000))
        parts.append(np.column_stack([centre + rng.normal(0, spread, size=(m, 2)), rng.normal(rng.uniform(-2.5, 0.5), rng.choice([0.0, 0.02, 0.4]), size=m)]))
    pts = np.concatenate(parts).astype(np.float32)
    if rng.integers(0, 2):
        rng.shuffle(pts)
    cloud = synth.make_cloud(pts, ring=rng.integers(0, 64, len(pts)))
    if knobs["pw"]:
        os.environ["GG_PW"] = str(knobs["pw"])
    else:
        os.environ.pop("GG_PW", None)
    try:
        seg = api.GroundSegmentation().init(length, resolution, n_slots=1, max_points=len(cloud))
        for key in ("front", "scan_parts", "sweep_gpw", "k2_per_cloud", "sweep_pair", "sweep_pair_wgs", "sweep_pair_waves"):
            if knobs[key]:
                seg.debug_set_tuning(key, knobs[key])
        seg.set_flags(minimal_layers=knobs["minimal"])
        ref = oracle.OracleMap(length, resolution)
        base_z = float(rng.uniform(-2.0, -1.4))
        for f in range(3):
            out, labels, index = seg.filter_cloud(cloud, t.ORIGIN0, base_z, return_details=True)
            r = ref.filter_cloud(cloud, t.ORIGIN0, base_z)
            assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), f"frame {f}: labels / order"
            assert out.tobytes() == r["out_points"].tobytes(), f"frame {f}: returned cloud"
            if f != 1:  # (frame 1 leaves the lazily computed layers owed while the next cloud arrives)
                t.assert_same_state(seg.map(0), ref, f"frame {f}")
        seg.close()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("seed", seed, knobs, (length, resolution), "FAILED:", str(e)[:300], flush=True)
print("done, seeds", first, "..", last - 1, "failures:", bad)
