"""What the two unpinned third-party conventions change: run the oracle under each setting and count.

  * Eigen 3.3 (redux_novec_unroller) vs Eigen 3.4 + SSE (slice-vectorised) order of the 5x5 block sums (:359, :374-375)
    -- on the committed vector tests/golden/hdl64_small_364.npz and on a full-size HDL-64E cloud, several frames;
  * KDL::Rotation::Quaternion vs tf2::Matrix3x3::setRotation in doTransform(PointStamped) (src/GroundGrid.cpp:129)
    -- on a driving sequence whose tf quaternions are unit only up to rounding, as a tf tree delivers them.

    python tools/convention_blast_radius.py        (CPU only; prints a JSON object)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def run(cloud, frames, length=120.0, res=0.33, pos=(0.0, 0.0), origin=(0.0, 0.0, 0.0), base_z=-1.73):
    m = oracle.OracleMap(length, res, pos=pos)
    out = []
    for _ in range(frames):
        r = m.filter_cloud(cloud, origin, base_z)
        out.append((r["label"].copy(), m.layer("ground").copy(), m.layer("groundpatch").copy()))
    return out


def diff(a, b):
    rows = []
    for f, ((la, ga, pa), (lb, gb, pb)) in enumerate(zip(a, b)):
        dg = np.abs(ga.astype(np.float64) - gb.astype(np.float64))
        rows.append({"frame": f, "labels_differ": int((la != lb).sum()), "of_points": int(la.size),
                     "ground_cells_differ": int((ga != gb).sum()), "max_abs_dground_m": float(np.nanmax(dg)) if dg.size else 0.0,
                     "groundpatch_cells_differ": int((pa != pb).sum())})
    return rows


def eigen_case(cloud, frames, **kw):
    oracle.set_eigen_reduction(0)
    a = run(cloud, frames, **kw)
    oracle.set_eigen_reduction(1)
    b = run(cloud, frames, **kw)
    oracle.set_eigen_reduction(0)
    return diff(a, b)


def rotation_case(n_frames=12):
    base = synth.hdl64_cloud(seed=77, n_az=500)
    res = {}
    for conv in ("kdl", "tf2"):
        m = oracle.OracleMap(120.0, 0.33)
        rng = np.random.default_rng(5)
        out = []
        for f in range(n_frames):
            yaw, pitch = 0.15 * f, 0.02 * np.sin(0.7 * f)
            x, y = 1.3 * f, 0.135 * f * f - 0.7 * f
            c, s = np.float32(np.cos(yaw)), np.float32(np.sin(yaw))
            cloud = synth.clone_cloud(base)
            cloud["x"] = (c * base["x"] - s * base["y"] + np.float32(x)).astype(np.float32)
            cloud["y"] = (s * base["x"] + c * base["y"] + np.float32(y)).astype(np.float32)
            # base_link <- map: a quaternion composed like tf does (product of two rotations), unit only up to rounding
            q = np.array([0.0, np.sin(pitch / 2), 0.0, np.cos(pitch / 2)]) * (1.0 + rng.uniform(-2e-16, 2e-16))
            qz = np.array([0.0, 0.0, np.sin(-yaw / 2), np.cos(-yaw / 2)])
            qq = np.array([q[3] * qz[0] + q[0] * qz[3] + q[1] * qz[2] - q[2] * qz[1], q[3] * qz[1] - q[0] * qz[2] + q[1] * qz[3] + q[2] * qz[0],
                           q[3] * qz[2] + q[0] * qz[1] - q[1] * qz[0] + q[2] * qz[3], q[3] * qz[3] - q[0] * qz[0] - q[1] * qz[1] - q[2] * qz[2]])
            if f:
                m.update(x, y, (-x, -y, 1.73, qq[0], qq[1], qq[2], qq[3]), rotation=conv)
            r = m.filter_cloud(cloud, (x, y, 0.0), -1.73)
            out.append((r["label"].copy(), m.layer("ground").copy(), m.layer("groundpatch").copy()))
        res[conv] = out
    return diff(res["kdl"], res["tf2"])


if __name__ == "__main__":
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hdl64_small_364.npz"))
    small = np.frombuffer(g["cloud"].tobytes(), dtype=synth.POINT_DTYPE)
    result = {
        "eigen33_vs_eigen34sse": {
            "hdl64_small_364 (committed vector, 2 frames)": eigen_case(small, int(g["frames"]), origin=tuple(g["origin"]), base_z=float(g["base_z"])),
            "hdl64 full size, seed 20240113, 5 frames": eigen_case(synth.hdl64_cloud(), 5),
        },
        "kdl_vs_tf2_rotation (12-frame drive, quaternions unit up to rounding)": rotation_case(),
    }
    print(json.dumps(result, indent=1))
