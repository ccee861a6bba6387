"""N3 (SemanticKITTI I/O + evaluator) on CPU: file formats, pose math, frame wiring, the evaluator pinned against the
result table the reference publishes (README.md:57-94, tests/golden/readme_seq00_table.json), and BASELINE configs[0]
plumbing: a KITTI-format sequence through the CPU oracle end to end."""
import json
import os

import numpy as np
import pytest

from groundgrid_amd import kitti, replay, synth
from groundgrid_amd.evaluate import GroundEvaluator, LABELS
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "readme_seq00_table.json")


def test_evaluator_reproduces_the_published_seq00_summary():
    g = json.load(open(GOLDEN))
    ev = GroundEvaluator.from_counts({k: v["nonground"] for k, v in g["labels"].items()}, {k: v["total"] for k, v in g["labels"].items()})
    s = ev.summary()
    S = g["summary"]
    assert (s["TP"], s["FP"]) == (S["Precision"][1], S["Precision"][2]) == (209090638, 8607231)
    assert s["FN"] == S["Recall"][2] == 2761917
    assert s["TP"] + s["TN"] == S["Accuracy"][1] and s["TP"] + s["TN"] + s["FP"] + s["FN"] == S["Accuracy"][2]
    for key, name in (("precision", "Precision"), ("recall", "Recall"), ("f1", "F1"), ("accuracy", "Accuracy"), ("iou_ground", "IoUg")):
        assert f"{s[key]:2.2%}" == f"{S[name][0]:.2f}%", key
    # per-label percentages as printed
    txt = ev.table()
    for name, row in g["labels"].items():
        assert f"{row['nonground'] / row['total']:2.2%}" == f"{row['nonground_pct']:.2f}%"
        assert str(row["total"]) in txt


def test_evaluator_counts_points_like_the_reference_callback():
    ev = GroundEvaluator()
    sem = np.array([40, 40, 40, 10, 10, 70, 72, 0, 48], dtype=np.uint16)   # road x3, car x2, vegetation, terrain, unlabeled, sidewalk
    pred = np.array([49, 49, 99, 99, 49, 49, 49, 99, 49], dtype=np.uint8)
    ev.add_cloud(pred, sem)
    assert ev.true_positive["road"] == 2 and ev.non_ground["road"] == 1 and ev.total["road"] == 3
    assert ev.false_positive["car"] == 1 and ev.non_ground["car"] == 1
    assert ev.false_positive["vegetation"] == 1        # counted, but vegetation is in none of the three lists
    s = ev.summary()
    assert (s["TP"], s["FP"], s["FN"], s["TN"]) == (4, 1, 1, 1)
    with pytest.raises(KeyError):
        ev.add_cloud(np.array([49]), np.array([7]))    # label id outside the yaml: the reference raises too


def _synthetic_sequence(tmp_path, n_frames=4, n_az=260):
    base = synth.hdl64_cloud(seed=5, n_az=n_az)
    # fake semantic labels: ground-ish points "road", high points "building", a few "vegetation"
    lab = np.where(base["z"] < -1.4, 40, 50).astype(np.uint16)
    lab[::17] = 70
    base["ring"] = lab
    calib = np.vstack((np.array(kitti.CALIB_STRING.split(), dtype=np.float64).reshape(3, 4), [0, 0, 0, 1]))
    poses_cam = []
    for f in range(n_frames):
        yaw, x, y = 0.1 * f, 1.1 * f, 0.2 * f
        T = np.eye(4)
        T[:3, :3] = [[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]]
        T[:3, 3] = [x, y, 0.01 * f]
        poses_cam.append(calib @ T @ np.linalg.inv(calib))  # so that calib^-1 . P . calib == T
    d = str(tmp_path / "seq")
    kitti.write_synthetic_sequence(d, [base] * n_frames, poses_cam)
    return d, base


def test_reader_pose_math_and_frame_wiring(tmp_path):
    d, base = _synthetic_sequence(tmp_path)
    seq = kitti.KittiSequence(d)
    assert len(seq) == 4 and seq.have_labels
    fr = seq.frame(2)
    assert np.array_equal(fr.cloud_sensor["x"], base["x"]) and np.array_equal(fr.cloud_sensor["ring"], base["ring"])
    assert np.array_equal(fr.cloud_sensor["intensity"], base["intensity"])
    # pose = calib^-1 . P . calib recovers the vehicle pose we wrote
    yaw, x, y = 0.2, 2.2, 0.4
    assert np.allclose(seq.poses[2][:3, 3], [x, y, 0.02], atol=1e-9)
    assert np.allclose(seq.poses[2][:2, :2], [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]], atol=1e-9)
    # cloud in the map frame = R p + t in double, cast to float (Nodelet.cpp:166-181)
    ex = (np.cos(yaw) * base["x"].astype(np.float64) - np.sin(yaw) * base["y"].astype(np.float64) + x)
    assert np.max(np.abs(fr.cloud_map["x"] - ex)) < 1e-5
    assert np.allclose(fr.origin, (x, y, 0.02), atol=1e-6) and np.allclose(fr.odom, (x, y, 0.02))
    # map <- base_link = pose . (1.95, 0, -1.73): only translation.z reaches the path
    assert abs(fr.map_to_base_z - (0.02 - 1.73)) < 1e-9
    # base_link <- map really is the inverse: it maps the base_link origin (in map coordinates) to 0
    R = kitti.matrix_from_quaternion(fr.base_to_map[3:])
    base_in_map = np.array([x + 1.95 * np.cos(yaw), y + 1.95 * np.sin(yaw), 0.02 - 1.73])
    assert np.allclose(R @ base_in_map + np.array(fr.base_to_map[:3]), 0, atol=1e-9)
    # quaternion <-> matrix helpers are consistent with each other
    q = kitti.quaternion_from_matrix(seq.poses[2])
    assert np.allclose(kitti.matrix_from_quaternion(q), seq.poses[2][:3, :3], atol=1e-12)


class OracleBackend:
    """BASELINE configs[0]: the CPU reference path behind the same replay harness (plumbing, no GPU)."""

    def __init__(self):
        self.m = None

    def reset(self, pos, odom_z):
        self.m = oracle.OracleMap(120.0, 0.33, pos=pos, odom_z=float(odom_z))

    def move(self, odom, base_to_map):
        self.m.update(odom[0], odom[1], base_to_map)

    def filter(self, cloud_map, origin, base_z):
        r = self.m.filter_cloud(cloud_map, origin, base_z)
        return r["label"], r["index"]


def test_config0_cpu_plumbing_sequence_through_the_oracle(tmp_path):
    d, _ = _synthetic_sequence(tmp_path)
    seq = kitti.KittiSequence(d)
    ev, spent = replay.replay(seq, OracleBackend())
    s = ev.summary()
    assert ev.cloud_count == 4 and spent > 0
    assert sum(ev.total.values()) > 4 * 10000
    assert s["TP"] > 0 and s["TN"] > 0 and 0.5 < s["accuracy"] <= 1.0   # "road" points are found as ground, "building" as obstacles
    assert "Precision" in ev.table() and set(LABELS.values()) >= {k for k, v in ev.total.items() if v}


def test_compare_with_the_published_table():
    """bench.py --kitti-dir diffs the replay's table against README.md:57-94: identical counts -> zero deltas, within tolerance;
    a shifted label falls out of it."""
    g = json.load(open(GOLDEN))
    ev = GroundEvaluator.from_counts({k: v["nonground"] for k, v in g["labels"].items()}, {k: v["total"] for k, v in g["labels"].items()})
    ev.cloud_count = g["clouds"]
    d = ev.compare_with(g, tolerance_pct=0.01)
    assert d["within_tolerance"] and all(abs(v["delta"]) <= 0.005 for v in d["summary"].values())
    assert all(abs(v["nonground_pct_delta"]) <= 0.005 and v["total_rel_pct"] == 0 for v in d["labels"].values())
    ev.non_ground["car"] -= 2_000_000
    ev.false_positive["car"] += 2_000_000
    d = ev.compare_with(g, tolerance_pct=1.0)
    assert not d["within_tolerance"] and d["labels"]["car"]["nonground_pct_delta"] < -4


def test_player_euler_round_trip_of_the_broadcast_quaternion():
    """kitti_data_publisher.py:208-214 broadcasts quaternion_from_euler(*euler_from_quaternion(q)): the same rotation up to
    rounding (and never far from q), identity for the identity, and frames built with the flag differ in the last ulps only."""
    rng = np.random.default_rng(5)
    assert np.array_equal(kitti.euler_roundtrip_quaternion([0, 0, 0, 1.0]), [0, 0, 0, 1.0])
    for _ in range(200):
        ang = rng.uniform(-3.1, 3.1)
        tilt = rng.normal(0, 0.05, 2)
        q = np.array([tilt[0], tilt[1], np.sin(ang / 2), np.cos(ang / 2)])
        q /= np.linalg.norm(q)
        q2 = kitti.euler_roundtrip_quaternion(q)
        assert np.max(np.abs(q2 - q)) < 1e-14 or np.max(np.abs(q2 + q)) < 1e-14
    cloud = synth.hdl64_cloud(seed=3, n_az=64)
    pose = np.eye(4)
    c, s_ = np.cos(0.3), np.sin(0.3)
    pose[:2, :2] = [[c, -s_], [s_, c]]
    pose[:3, 3] = [10.0, -4.0, 0.2]
    f0 = kitti.make_frame(0, cloud, pose)
    f1 = kitti.make_frame(0, cloud, pose, euler_roundtrip=True)
    assert np.max(np.abs(f0.cloud_map["x"] - f1.cloud_map["x"])) < 1e-4


def test_synthetic_drive_is_a_closed_loop_wired_like_a_sequence():
    """groundgrid_amd.kitti.synthetic_drive (the configs[4]-shaped leg of bench.py): frames in memory, wired by make_frame like a
    sequence directory; 0.8 m per frame, back at the start after the last one; through the CPU path (plumbing, no GPU)."""
    poses = kitti.drive_poses(300)
    steps = [np.linalg.norm(poses[i + 1][:2, 3] - poses[i][:2, 3]) for i in range(299)]
    assert 0.7 < min(steps) and max(steps) < 0.9
    assert np.linalg.norm(poses[-1][:2, 3] - poses[0][:2, 3]) < 1.0
    frames = list(kitti.synthetic_drive(6, n_scenes=2, n_az=200))
    assert [f.index for f in frames] == list(range(6)) and frames[0].cloud_sensor is frames[2].cloud_sensor  # two scenes in turn
    assert abs(frames[3].map_to_base_z - (kitti.drive_poses(6)[3][2, 3] - 1.73)) < 1e-9  # (a yaw-only pose: base_link sits 1.73 m below)
    ev, _ = replay.replay(frames, OracleBackend())
    assert ev.cloud_count == 6 and sum(ev.total.values()) > 0
