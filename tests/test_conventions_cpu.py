"""The third-party conventions the reference inherits (unpinned: tools/pin/ decides them on a ROS box) are single
swappable functions.  These CPU tests hold every implementation of each variant -- the library's host helper, the C
oracle, the independent Python restatement -- to the same bits, show where the variants differ, and pin the Eigen 3.4
order to a lane-by-lane emulation of the SSE2 reduction."""
import ctypes as C

import numpy as np
import pytest

from groundgrid_amd import _lib, api, build, kitti
from oracle import oracle
from tests import pyref

F32 = np.float32


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def lib_rotation(lib, q, conv):
    qq = (C.c_double * 4)(*q)
    R = (C.c_double * 9)()
    assert lib.gg_rotation_from_quaternion(conv, qq, R) == 0
    return np.array(list(R)).reshape(3, 3)


def test_rotation_conventions_agree_across_implementations(lib):
    rng = np.random.default_rng(11)
    differ = 0
    for k in range(300):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        if k % 3 == 0:
            q *= 1.0 + rng.uniform(-1e-7, 1e-7)  # the message quaternion is normalised in double at best
        for name, conv in (("tf2", 0), ("kdl", 1)):
            a = lib_rotation(lib, q, conv)
            b = oracle.rotation_from_quaternion(q, name)
            c = kitti.matrix_from_quaternion(q, name)
            assert a.tobytes() == b.tobytes() == c.tobytes(), (name, q)
            assert np.allclose(a @ a.T, np.eye(3), atol=1e-6)
        t, kd = lib_rotation(lib, q, 0), lib_rotation(lib, q, 1)
        assert np.allclose(t, kd, atol=1e-6)
        differ += t.tobytes() != kd.tobytes()
    assert differ > 100  # the two constructions round differently almost always: the choice is observable


def test_transform_from_pose_layout(lib):
    pose = (1.5, -2.0, 0.25, 0.1, -0.2, 0.3, 0.927)
    M = api.transform_from_pose(pose, "kdl")
    assert M.shape == (3, 4) and tuple(M[:, 3]) == pose[:3]
    assert M[:, :3].tobytes() == oracle.rotation_from_quaternion(pose[3:], "kdl").tobytes()
    assert oracle.matrix_from_pose(pose, "tf2").tobytes() == api.transform_from_pose(pose, "tf2").tobytes()
    assert oracle.plane_from_pose(pose, "kdl") == (M[2, 0], M[2, 1], M[2, 2], 0.25)


def test_kdl_quaternion_entries_by_hand():
    # KDL::Rotation::Quaternion(x, y, z, w): third row = (2xz - 2wy, 2yz + 2wx, w2 - x2 - y2 + z2)
    x, y, z, w = 0.1, -0.2, 0.3, 0.9
    R = oracle.rotation_from_quaternion((x, y, z, w), "kdl")
    assert R[2, 0] == 2 * x * z - 2 * w * y and R[2, 1] == 2 * y * z + 2 * w * x
    assert R[2, 2] == w * w - x * x - y * y + z * z
    assert R[0, 0] == w * w + x * x - y * y - z * z and R[0, 1] == 2 * x * y - 2 * w * z
    # tf2::Matrix3x3::setRotation normalises (s = 2 / |q|^2): a quaternion of length 2 gives the same rotation
    Rt = oracle.rotation_from_quaternion((x, y, z, w), "tf2")
    Rt2 = oracle.rotation_from_quaternion((2 * x, 2 * y, 2 * z, 2 * w), "tf2")
    assert np.allclose(Rt, Rt2, atol=1e-15)
    # ... KDL does not: it scales by |q|^2
    Rk2 = oracle.rotation_from_quaternion((2 * x, 2 * y, 2 * z, 2 * w), "kdl")
    assert np.allclose(Rk2, 4 * R, atol=1e-14)


def sse_sum_5x5(e):
    """Eigen 3.4 redux, SliceVectorizedTraversal, Packet4f, written as the SSE instructions it compiles to."""
    e = np.asarray(e, dtype=np.float32)
    acc = e[0:4].copy()                                   # packet(0, 0)
    for j in range(1, 5):
        acc = (acc + e[5 * j:5 * j + 4]).astype(np.float32)  # padd, column j rows 0..3
    tmp = (acc + np.array([acc[2], acc[3], acc[2], acc[3]], dtype=np.float32)).astype(np.float32)  # a + movehl(a, a)
    res = F32(tmp[0] + tmp[1])                            # add_ss(tmp, shuffle(tmp, 1))
    for j in range(5):
        res = F32(res + e[5 * j + 4])                     # scalar tail: row 4 of every column
    return res


def test_eigen34_block_sum_order():
    rng = np.random.default_rng(2)
    differ = 0
    try:
        for _ in range(400):
            v = (rng.standard_normal(25) * 10.0 ** rng.integers(-4, 5, size=25)).astype(np.float32)
            ptr = v.ctypes.data_as(C.POINTER(C.c_float))
            oracle.set_eigen_reduction(0)
            pyref.EIGEN_REDUCTION = 0
            a33 = oracle.lib().ggo_block_sum(ptr, 5)
            assert a33 == oracle.tree_sum(v) == float(pyref.block_sum(list(v)))
            oracle.set_eigen_reduction(1)
            pyref.EIGEN_REDUCTION = 1
            a34 = oracle.lib().ggo_block_sum(ptr, 5)
            assert a34 == float(sse_sum_5x5(v)) == float(pyref.block_sum(list(v)))
            differ += a33 != a34
            w = v[:9].copy()  # 3x3 blocks take the unrolled tree under both
            assert oracle.lib().ggo_block_sum(w.ctypes.data_as(C.POINTER(C.c_float)), 3) == oracle.tree_sum(w)
    finally:
        oracle.set_eigen_reduction(0)
        pyref.EIGEN_REDUCTION = 0
    assert differ > 200


def test_eigen34_changes_patch_results_and_restatements_still_agree():
    """Whole path under the Eigen 3.4 order: C oracle == independent Python restatement, and the terrain differs from the
    Eigen 3.3 result in some cell (the convention is observable, so it must be pinned)."""
    rng = np.random.default_rng(4)
    n = 9000
    xy = rng.uniform(-10, 10, size=(n, 2))
    z = -1.7 + 0.05 * xy[:, 0] + rng.normal(0, 0.03, size=n)
    cloud = oracle.make_cloud(np.column_stack([xy, z]).astype(np.float32), ring=rng.integers(0, 64, n))

    def run(order):
        oracle.set_eigen_reduction(order)
        pyref.EIGEN_REDUCTION = order
        m = oracle.OracleMap(21.12, 0.33)
        p = pyref.PyRef(21.12, 0.33)
        p.expected = m.expected_points().copy()
        cfg_far = 1.0  # patch_size_change_distance: make most of this small map use 5x5 patches
        m.cfg.patch_size_change_distance = cfg_far
        p.cfg["patch_size_change_distance"] = cfg_far
        for _ in range(2):
            r = m.filter_cloud(cloud, (0, 0, 0), -1.7)
            q = p.filter_cloud(cloud, (0, 0, 0), -1.7)
        for name in oracle.LAYERS:
            assert np.array_equal(m.layer(name), p.L[name], equal_nan=True), (order, name)
        assert np.array_equal(r["label"], q["label"])
        return m.layer("ground").copy()

    try:
        g33, g34 = run(0), run(1)
    finally:
        oracle.set_eigen_reduction(0)
        pyref.EIGEN_REDUCTION = 0
    assert np.max(np.abs(g33 - g34)) < 1e-4 and not np.array_equal(g33, g34)


def test_line_of_sight_walk_bound_is_shared():
    """A corrupt z far below the map: the walk is cut at GGO_WALK_MAX_STEP steps in the oracle like in the library
    (documented deviation) and returns promptly."""
    import time

    m = oracle.OracleMap(21.12, 0.33)
    good = np.column_stack([np.random.default_rng(0).uniform(-8, 8, size=(4000, 2)), np.full(4000, -1.7)]).astype(np.float32)
    cloud = oracle.make_cloud(good, ring=np.zeros(4000, dtype=np.int64))
    for _ in range(3):
        m.filter_cloud(cloud, (0, 0, 0), -1.7)  # warm map: confidences above 0.01
    bad = oracle.make_cloud(np.array([[4.0, 4.0, -1e9], [4.0, 4.0, -3e38], [0.2, 5.0, -1e7]], dtype=np.float32), ring=np.zeros(3, dtype=np.int64))
    t0 = time.perf_counter()
    r = m.filter_cloud(bad, (0, 0, 0), -1.7)
    assert time.perf_counter() - t0 < 5.0
    assert set(r["cls"].tolist()) <= {oracle.KEPT, oracle.OUTLIER}
