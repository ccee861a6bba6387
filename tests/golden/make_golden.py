"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/gg_oracle.c).

These are REGRESSION vectors of this repository's own oracle on seeded synthetic inputs, not outputs of the
reference: the reference ships no tests or vectors and cannot be built or run in this image
(parity unpinned).  They (a) pin the oracle against accidental change and (b) travel to
the GPU box, where the HIP path is compared with them bit for bit.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from groundgrid_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def save(name, cloud, length, resolution, pos, origin, base_z, frames):
    m = oracle.OracleMap(length, resolution, pos=pos)
    d = dict(cloud=np.frombuffer(cloud.tobytes(), dtype=np.uint8), length=length, resolution=resolution,
             pos=np.array(pos, dtype=np.float64), origin=np.array(origin, dtype=np.float32), base_z=base_z, frames=frames)
    for f in range(frames):
        r = m.filter_cloud(cloud, origin, base_z)
        d[f"label_{f}"] = r["label"].copy()
        d[f"index_{f}"] = r["index"].copy()
        d[f"cls_{f}"] = r["cls"].copy()
        for layer in ("ground", "groundpatch", "variance"):
            d[f"{layer}_{f}"] = m.layer(layer).copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, len(cloud), "points;", {int(k): int(v) for k, v in zip(*np.unique(r["cls"], return_counts=True))})


def micro_cloud(seed, n, pos):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-12, 12, size=(n, 2)) + np.array(pos)
    z = -1.7 + 0.03 * xy[:, 0] + rng.normal(0, 0.02, size=n)
    z[rng.random(n) < 0.2] += rng.uniform(0.2, 2.0)
    z[rng.random(n) < 0.03] -= 1.0
    return oracle.make_cloud(np.column_stack([xy, z]).astype(np.float32), ring=rng.integers(0, 64, n))


if __name__ == "__main__":
    # 1. 364 x 364, ~5 k HDL-64E points, 2 frames
    save("hdl64_small_364", synth.hdl64_cloud(seed=20240113, n_az=84), 120.0, 0.33, (0.0, 0.0), (0.0, 0.0, 0.0), -1.73, 2)
    # 2. 64 x 64 micro grid, shifted map and sensor origin, 3 frames (stateful; outlier ray march fires)
    save("micro_64_stateful", micro_cloud(1, 12000, (1.3, -2.1)), 21.12, 0.33, (1.3, -2.1), (0.4, -0.3, 0.2), -1.7, 3)
    # 3. edge cases: outside, border rows, ring > max_ring (default 1024 never; use ring 2000), NaN, inside sqrt(12) m
    pts = np.array([[5, 5, -1], [1, 1, -1], [5, 5, -1], [50, 0, -1], [np.nan, 0, -1], [5, 5, np.nan], [-10.3, -10.3, -1],
                    [5.1, 5.0, -1.2], [0, 0, 3], [-10.5, 3, -1.6], [3, -10.5, -1.6]], dtype=np.float32)
    save("edge_cases_64", oracle.make_cloud(pts, ring=[0, 0, 2000, 0, 0, 0, 0, 1, 2, 3, 4]), 21.12, 0.33, (0.0, 0.0), (0.0, 0.0, 0.0), -1.7, 2)
    # 4. unstructured cloud on the full grid, moved map
    save("random_364_moved", synth.random_cloud(6000, seed=3), 120.0, 0.33, (4.29, -2.64), (4.0, -2.5, 0.1), -1.6, 2)
