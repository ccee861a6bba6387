"""CPU tests of the oracle (oracle/gg_oracle.c): hand-computed cases, an independent pure-Python
restatement (tests/pyref.py) on small grids, the committed regression vectors (tests/golden/), and the
edge cases the reference's control flow has.  The reference itself ships no tests or vectors
 and cannot be built here, so nothing below is a reference output: PARITY UNPINNED.
"""
import math
import os

import numpy as np
import pytest

from groundgrid_amd import synth
from oracle import oracle
from tests import pyref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
F32 = np.float32


def nan_equal(a, b):
    return np.array_equal(a, b, equal_nan=True)


# ------------------------------------------------------------------ geometry / tables

def test_geometry_matches_reference_constants():
    m = oracle.OracleMap(120.0, 0.33)
    assert (m.rows, m.cols) == (364, 364)  # round(120 / 0.33000001311) = 364, not 363
    assert m.resolution == float(F32(0.33))
    assert m.length[0] == 364 * float(F32(0.33))
    assert oracle.spiral_visit_count(364) == 130680
    assert oracle.spiral_visit_count(1000) == 995004


def test_inconsistent_geometry_rejected():
    # grid_map size round(100.4/0.4)=251 but init() sees size_t(100)/0.4 -> 250
    with pytest.raises(ValueError):
        oracle.OracleMap(100.4, 0.4)


def test_get_index_against_plain_double_arithmetic():
    m = oracle.OracleMap(120.0, 0.33, pos=(3.7, -11.2))
    res, L = m.resolution, m.length[0]
    rng = np.random.default_rng(0)
    for x, y in rng.uniform(-80, 80, size=(2000, 2)):
        inside, r, c = m.get_index(x, y)
        half = 0.5 * L
        er = int(-(((x - half) - 3.7) / res))
        ec = int(-(((y - half) - (-11.2)) / res))
        tx, ty = -((x - 3.7) - half), -((y + 11.2) - half)
        assert (r, c) == (er, ec)
        assert inside == (0.0 <= tx < L and 0.0 <= ty < L)
    # index 0 is at +x/+y; NaN is never inside
    assert m.get_index(3.7 + 60.0, -11.2 + 60.0)[1:] == (0, 0)
    assert m.get_index(float("nan"), 0.0)[0] is False


def test_expected_points_table():
    m = oracle.OracleMap(120.0, 0.33)
    e = m.expected_points()
    vpad = F32(0.00174532925 * 2)
    assert e[182, 182] == F32(np.arctan(F32(np.inf), dtype=np.float32)) / vpad  # dist 0 -> atan(inf)/vpad ~ 450
    d = F32(math.hypot(10 - 182.0, 300 - 182.0))
    assert abs(e[10, 300] - F32(math.atan(1.0 / float(d))) / vpad) <= 2 * np.spacing(e[10, 300])
    assert e[0, 0] == e[0, 0].astype(np.float32) and np.isfinite(e).all()


def test_tree_sum_order_is_eigen_unrolled_tree():
    rng = np.random.default_rng(1)
    for n in (9, 25):
        for _ in range(50):
            v = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, size=n)).astype(np.float32)
            assert oracle.tree_sum(v) == float(pyref.tree_sum(list(v)))
    # order sensitivity: a left-to-right sum gives a different float here
    v = np.array([1e8, 1.0, -1e8, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], dtype=np.float32)
    lr = F32(0)
    for x in v:
        lr = F32(lr + x)
    t = ((F32(v[0] + v[1]) + F32(v[2] + v[3])) + (F32(v[4] + v[5]) + F32(v[6] + F32(v[7] + v[8]))))
    assert oracle.tree_sum(v) == float(F32(t)) and float(lr) != float(F32(t))


def test_hypotf_is_double_sqrt_rounded():
    rng = np.random.default_rng(2)
    for x, y in rng.uniform(-100, 100, size=(500, 2)).astype(np.float32):
        assert oracle.hypotf(x, y) == float(F32(math.sqrt(float(x) * float(x) + float(y) * float(y))))
    assert oracle.hypotf(0.0, -3.0) == 3.0


def test_double_rounding_identity():
    """(float)((double)a / (double)b) == a / b in binary32 for float a, b (53 >= 2*24+2): lets K2 use f32 division."""
    rng = np.random.default_rng(7)
    a = (rng.standard_normal(2_000_000) * 10.0 ** rng.integers(-3, 4, size=2_000_000)).astype(np.float32)
    b = rng.integers(1, 5000, size=2_000_000).astype(np.float32)
    via_double = (a.astype(np.float64) / b.astype(np.float64)).astype(np.float32)
    assert np.array_equal(via_double, a / b)
    b2 = (rng.random(2_000_000) * 100 + 0.01).astype(np.float32)
    assert np.array_equal((a.astype(np.float64) / b2.astype(np.float64)).astype(np.float32), a / b2)


# ------------------------------------------------------------------ hand-computed reductions

def small_map():
    return oracle.OracleMap(21.12, 0.33)  # 64 x 64 cells


def test_single_cell_welford_by_hand():
    m = small_map()
    assert m.rows == 64
    # three points in the same cell (5.0, 5.0), sensor origin z = 0.5
    zs = np.array([-1.0, -0.8, -1.3], dtype=np.float32)
    xyz = np.column_stack([np.full(3, 5.0), np.full(3, 5.0), zs]).astype(np.float32)
    cloud = oracle.make_cloud(xyz, ring=[1, 2, 3])
    m.stage_reset()
    cls, cell = m.stage_insert(cloud, origin=(0.0, 0.0, 0.5))
    assert (cls == oracle.KEPT).all() and len(set(cell)) == 1
    r, c = int(cell[0]) % 64, int(cell[0]) // 64
    inside, er, ec = m.get_index(5.0, 5.0)
    assert inside and (r, c) == (er, ec)
    # by hand (float32 steps of src/GroundSegmentation.cpp:295-309)
    cnt, gc, mean, pdm, m2 = F32(0), F32(0), F32(0), F32(0), F32(0)
    mx, mn = np.finfo(np.float32).tiny, np.finfo(np.float32).max
    for z in zs:
        pd = F32(z - F32(0.5))
        gc = F32(np.float64(F32(z + F32(cnt * gc))) / (np.float64(cnt) + 1.0))
        if mean == 0:
            mean = pd
        delta = F32(pd - mean)
        mean = F32(mean + F32(delta / F32(cnt + F32(1))))
        pdm = F32(np.float64(F32(pd + F32(cnt * pdm))) / (np.float64(cnt) + 1.0))
        m2 = F32(m2 + F32(delta * F32(pd - mean)))
        mx = max(mx, z)
        mn = min(mn, F32(z - F32(0.0001)))
        cnt = F32(cnt + 1)
    assert m.layer("points")[r, c] == 3 and m.layer("pointsRaw")[r, c] == 3
    assert m.layer("groundCandidates")[r, c] == gc
    assert m.layer("meanVariance")[r, c] == mean
    assert m.layer("planeDist")[r, c] == pdm
    assert m.layer("m2")[r, c] == m2
    assert m.layer("maxGroundHeight")[r, c] == mx
    assert m.layer("minGroundHeight")[r, c] == mn
    # untouched cells hold the reset values, including the reference's FLT_MIN (sic) for max
    assert m.layer("minGroundHeight")[0, 0] == np.finfo(np.float32).max
    assert m.layer("maxGroundHeight")[0, 0] == np.finfo(np.float32).tiny


def test_welford_is_order_dependent():
    """m2 depends on point order -> unordered atomics cannot reproduce it."""
    rng = np.random.default_rng(5)
    zs = rng.normal(-1.7, 0.05, size=40).astype(np.float32)
    res = []
    for order in (np.arange(40), np.arange(40)[::-1]):
        m = small_map()
        xyz = np.column_stack([np.full(40, 5.0), np.full(40, 5.0), zs[order]]).astype(np.float32)
        m.stage_reset()
        _, cell = m.stage_insert(oracle.make_cloud(xyz), origin=(0, 0, 0))
        res.append(m.layer("m2").ravel(order="F")[cell[0]])
    assert res[0] != res[1]
    assert abs(res[0] - res[1]) < 1e-4


def test_classification_rules():
    m = small_map()
    cfg = m.cfg
    cfg.max_ring = 10
    pts = np.array([
        [5.0, 5.0, -1.0],     # kept
        [1.0, 1.0, -1.0],     # sqdist 2 < 12 -> ignored
        [5.0, 5.0, -1.0],     # ring 11 > max_ring -> ignored
        [50.0, 0.0, -1.0],    # outside the 21 m map
        [np.nan, 0.0, -1.0],  # NaN x -> outside
        [5.0, 5.0, np.nan],   # NaN z is KEPT (no comparison is true), poisons the cell
        [-10.3, -10.3, -1.0], # last rows/cols: kept by insert, dropped from the returned cloud (:167-168)
    ], dtype=np.float32)
    cloud = oracle.make_cloud(pts, ring=[0, 0, 11, 0, 0, 0, 0])
    r = m.filter_cloud(cloud, (0, 0, 0), -1.7)
    assert list(r["cls"]) == [3, 1, 1, 0, 0, 3, 3]
    assert list(r["cell"][[3, 4]]) == [-1, -1]
    assert r["label"][3] == 0 and r["label"][4] == 0 and r["index"][3] == -1
    inside, row, col = m.get_index(-10.3, -10.3)
    assert inside and (64 <= row + 3 or 64 <= col + 3)
    assert r["label"][6] == 0 and r["index"][6] == -1
    # order of the returned cloud: kept (cloud order) then ignored (cloud order)
    assert list(r["index"][[0, 5, 1, 2]]) == [0, 1, 2, 3]
    assert len(r["out_points"]) == 4
    assert set(np.unique(r["out_points"]["intensity"])) <= {49.0, 99.0}
    # NaN z: groundCandidates of that cell is NaN, points counted
    rr, cc = int(r["cell"][0]) % 64, int(r["cell"][0]) // 64
    assert np.isnan(m.layer("groundCandidates")[rr, cc])


def test_empty_cloud_and_all_outside():
    m = small_map()
    r = m.filter_cloud(synth.empty_cloud(0), (0, 0, 0), -1.0)
    assert len(r["out_points"]) == 0
    c = m.n = m.rows // 2 - 1
    assert m.layer("ground")[c, c] == F32(-1.0) and m.layer("groundpatch")[c, c] == 1.0
    xyz = np.array([[500, 500, 0], [-500, 2, 0]], dtype=np.float32)
    r = m.filter_cloud(oracle.make_cloud(xyz), (0, 0, 0), -1.0)
    assert len(r["out_points"]) == 0 and (r["label"] == 0).all()


def test_variance_zero_and_dist_zero_tolerance_branches():
    """:171: variance 0 -> inf -> clamp 0.3; dist 0 and variance 0 -> NaN -> classified ground."""
    m = small_map()
    # one kept point far away: its cell has variance 0 -> tolerance 0.3
    xyz = np.array([[6.0, 0.0, 5.0]], dtype=np.float32)
    r = m.filter_cloud(oracle.make_cloud(xyz), (0, 0, 0), 0.0)
    assert r["label"][0] in (49, 99)
    # a point exactly at the origin (ignored: sqdist 0 < 12) with variance 0 in its cell and dist 0 -> NaN -> ground
    xyz = np.array([[0.0, 0.0, 50.0]], dtype=np.float32)
    m2 = small_map()
    r = m2.filter_cloud(oracle.make_cloud(xyz), (0, 0, 0), 0.0)
    assert r["cls"][0] == oracle.IGNORED and r["label"][0] == 49


# ------------------------------------------------------------------ independent restatement

@pytest.mark.parametrize("seed,origin,pos", [(0, (0.0, 0.0, 0.0), (0.0, 0.0)), (1, (0.4, -0.3, 0.2), (1.3, -2.1))])
def test_oracle_matches_independent_python_restatement(seed, origin, pos):
    rng = np.random.default_rng(seed)
    n = 12000
    xy = rng.uniform(-12, 12, size=(n, 2)) + np.array(pos)
    z = -1.7 + 0.03 * xy[:, 0] + rng.normal(0, 0.02, size=n)
    z[rng.random(n) < 0.2] += rng.uniform(0.2, 2.0)
    z[rng.random(n) < 0.03] -= 1.0  # below ground: exercises the outlier ray march once the map is warm
    cloud = oracle.make_cloud(np.column_stack([xy, z]).astype(np.float32), ring=rng.integers(0, 64, n))
    m = oracle.OracleMap(21.12, 0.33, pos=pos)
    p = pyref.PyRef(21.12, 0.33, pos=pos)
    assert np.max(np.abs(p.expected - m.expected_points())) <= 1e-4 * np.max(p.expected)
    p.expected = m.expected_points().copy()  # libm atanf vs numpy arctan may differ by an ulp; share the table
    for frame in range(3):
        r = m.filter_cloud(cloud, origin, -1.7)
        q = p.filter_cloud(cloud, origin, -1.7)
        assert np.array_equal(r["cls"], q["cls"]), frame
        assert np.array_equal(r["cell"], q["cell"]), frame
        for name in oracle.LAYERS:
            assert nan_equal(m.layer(name), p.L[name]), (frame, name)
        assert np.array_equal(r["label"], q["label"]) and np.array_equal(r["index"], q["index"]), frame
    assert (r["cls"] == oracle.OUTLIER).sum() > 0  # the ray march fired


# ------------------------------------------------------------------ committed regression vectors

def golden_files():
    return sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz")) if os.path.isdir(GOLDEN) else []


@pytest.mark.parametrize("fname", golden_files())
def test_oracle_reproduces_committed_vectors(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    m = oracle.OracleMap(float(g["length"]), float(g["resolution"]), pos=tuple(g["pos"]))
    cloud = np.frombuffer(g["cloud"].tobytes(), dtype=oracle.POINT_DTYPE)
    for f in range(int(g["frames"])):
        r = m.filter_cloud(cloud, tuple(g["origin"]), float(g["base_z"]))
        assert np.array_equal(r["label"], g[f"label_{f}"])
        assert np.array_equal(r["index"], g[f"index_{f}"])
        assert nan_equal(m.layer("ground"), g[f"ground_{f}"])
        assert nan_equal(m.layer("groundpatch"), g[f"groundpatch_{f}"])
        assert nan_equal(m.layer("variance"), g[f"variance_{f}"])


def test_golden_vectors_exist():
    assert len(golden_files()) >= 3


# ------------------------------------------------------------------ whole-cloud properties

def test_full_size_cloud_properties():
    cloud = synth.hdl64_cloud(seed=11, n_az=520)
    m = oracle.OracleMap(120.0, 0.33)
    for _ in range(2):
        r = m.filter_cloud(cloud, (0, 0, 0), -1.73)
    lab, idx = r["label"], r["index"]
    emitted = idx >= 0
    assert (lab[emitted] != 0).all() and (lab[~emitted] == 0).all()
    assert sorted(idx[emitted]) == list(range(int(emitted.sum())))          # a permutation of the returned cloud
    kept = (r["cls"] == 3) & emitted
    assert (np.diff(idx[kept]) > 0).all()                                    # cloud order preserved inside a class
    # `points` layer now holds the non-ground count per cell (:176)
    cnt = np.bincount(r["cell"][(lab == 99)], minlength=364 * 364).reshape((364, 364), order="F")
    assert np.array_equal(m.layer("points"), cnt.astype(np.float32))
    # flat synthetic terrain: most returns are ground
    assert (lab == 49).sum() > (lab == 99).sum()


# ------------------------------------------------------------------ N1: GroundGrid::update

def test_map_update_semantics():
    m = oracle.OracleMap(21.12, 0.33)
    g = np.arange(64 * 64, dtype=np.float32).reshape((64, 64), order="F")
    m.set_layer("ground", g)
    m.set_layer("groundpatch", np.full((64, 64), 0.5, dtype=np.float32))
    res = m.resolution
    # less than half a cell: nothing moves, position unchanged (src/GroundGrid.cpp:135-137)
    moved, shift = m.update(0.16, -0.16, (0, 0, 1, 0, 0, 0, 1))
    assert not moved and shift == (0, 0) and m.position == (0.0, 0.0)
    # +0.7 m in x = 2.12 cells -> 2 cells; row index 0 is at +x, so two NEW rows appear at rows 0..1
    moved, shift = m.update(0.7, -0.34, (0.0, 0.0, 1.25, 0, 0, 0, 1))
    assert moved and shift == (-2, 1)
    assert m.position == (2 * res, -1 * res)                       # snapped to whole cells, not to the odometry
    G, W = m.layer("ground"), m.layer("groundpatch")
    assert np.array_equal(G[2:, :63], g[:62, 1:])                  # old content shifted by (+2 rows, -1 col)
    assert (G[:2, :] == np.float32(-1.25)).all() and (G[:, 63] == np.float32(-1.25)).all()   # -(z in base_link), identity rotation
    assert (W[:2, :] == 0).all() and (W[:, 63] == 0).all() and (W[2:, :63] == 0.5).all()
    assert np.isnan(m.layer("points")[:2, :]).all() and np.isnan(m.layer("minGroundHeight")[:, 63]).all()
    assert not np.isnan(m.layer("points")[2:, :63]).any()
    # a pitched base: the fill is a plane -(m20 x + m21 y + tz) over the cell centres
    q = (0.0, np.sin(0.05), 0.0, np.cos(0.05))
    m2 = oracle.OracleMap(21.12, 0.33)
    m2.update(5.0, 0.0, (0.2, 0.0, 1.0) + q)
    L = m2.length[0]
    xs = m2.position[0] + (0.5 * L - 0.5 * res) - res * np.arange(64)
    m20 = 2 * (q[0] * q[2] - q[3] * q[1])
    expect = -(m20 * xs[:15] + 1.0)
    assert np.allclose(m2.layer("ground")[:15, 10], expect, atol=1e-6)
    # a jump larger than the map: every cell is new
    m3 = oracle.OracleMap(21.12, 0.33)
    m3.update(500.0, 0.0, (0, 0, 2, 0, 0, 0, 1))
    assert (m3.layer("ground") == -2).all() and (m3.layer("groundpatch") == 0).all()


def test_decay_multiply_guard_is_exact():
    """sweep_core.h decayed_confidence computes (float)max(x - x/d, 0.001) (GroundSegmentation.cpp:463-464, double arithmetic) as
    x - x*(1/d) unless the low 29 bits of that double's significand lie within 64 ulps of 2^28 (a binary32 rounding boundary) --
    then, or for d < 1.25, the division decides.  Whenever the guard accepts, the result must equal the divide form bit for bit;
    the guard must be at least as strict as round 2's interval test (both ends of +-2^-48 convert alike)."""
    rng = np.random.default_rng(5)
    for d in (1.25, 1.3, 2.0, 3.0, 5.0, 7.3, 1e6):
        x = np.concatenate([rng.random(2_000_000, dtype=np.float32), np.float32(10.0) ** rng.uniform(-30, 3, 500_000).astype(np.float32),
                            np.array([0.0, 1.0, 0.5, 1e-7, 0.001, 0.00125, 0.0012500001], dtype=np.float32)]).astype(np.float64)
        exact = np.maximum(x - x / d, 0.001).astype(np.float32)
        t = x - x * (1.0 / d)
        low = t.view(np.uint64).astype(np.uint64) & np.uint64(0x1FFFFFFF)
        near = ((low - np.uint64(0x10000000 - 64)) & np.uint64(0xFFFFFFFF)) <= np.uint64(128)
        fast = np.maximum(t, 0.001).astype(np.float32)
        ok = ~near
        assert ok.mean() > 0.999
        assert np.array_equal(fast[ok], exact[ok])
        # wherever the interval test of round 2 could not decide and the value is above the floor, the new guard refuses as well
        lo = np.maximum(t * (1.0 - 2.0 ** -48), 0.001).astype(np.float32)
        hi = np.maximum(t * (1.0 + 2.0 ** -48), 0.001).astype(np.float32)
        assert not np.any((lo != hi) & ok)
    # adversarial: doubles constructed right next to binary32 midpoints, pushed through the guard as `t`
    m = (np.arange(1 << 12, dtype=np.uint64) << np.uint64(41)) | np.uint64(0x10000000)  # significands with low 29 bits = 2^28
    for delta in range(-80, 81, 8):
        bits = (np.uint64(0x3FB) << np.uint64(52)) | ((m + np.uint64(delta % (1 << 64))) & np.uint64((1 << 52) - 1))
        low = bits & np.uint64(0x1FFFFFFF)
        near = ((low - np.uint64(0x10000000 - 64)) & np.uint64(0xFFFFFFFF)) <= np.uint64(128)
        assert near.all() == (abs(delta) <= 64), delta


def test_stage_members_one_by_one_compose_to_the_stage():
    """The oracle's single-stage entries (the twins gg_run_stage is tested against on the GPU): the four quadrants of
    detect_ground_patches in any order are detect_ground_patches; a quadrant is its cells' detect_ground_patch<3|5> calls; the sweep is
    its interpolate_cell visits in the reference's order (src/GroundSegmentation.cpp:325-337, :413-439)."""
    rng = np.random.default_rng(41)
    xy = rng.uniform(-16, 16, size=(9000, 2))
    z = -1.7 + 0.03 * xy[:, 0] + rng.normal(0, 0.02, size=9000)
    z[rng.random(9000) < 0.2] += rng.uniform(0.2, 2.0)
    cloud = oracle.make_cloud(np.column_stack([xy, z]).astype(np.float32), ring=rng.integers(0, 64, 9000))
    a, b, c = (oracle.OracleMap(33.0, 0.33) for _ in range(3))
    for m in (a, b, c):
        m.filter_cloud(cloud, (0.0, 0.0, 0.0), -1.73)  # (gives every layer contents: counts, heights, m2, old terrain)
    n = a.rows
    a.stage_detect()
    for section in (3, 1, 0, 2):
        b.stage_detect_section(section)
    # c: cell by cell, as :329-337 does
    m2, pts = c.layer("m2"), c.layer("points")
    c.layer("variance")[...] = m2 / (pts + np.float32(np.finfo(np.float32).tiny))
    res = np.float32(0.33)
    for i in range(2, 2 * (n // 2) - 2):
        for j in range(2, n - 2):
            sqdist = np.float32(((i - n / 2.0) ** 2 + (j - n / 2.0) ** 2) * (float(res) * float(res)))
            c.detect_ground_patch(3 if float(sqdist) <= 400.0 else 5, i, j)
    for name in ("ground", "groundpatch", "variance"):
        assert np.array_equal(a.layer(name), b.layer(name), equal_nan=True), name
        assert np.array_equal(a.layer(name), c.layer(name), equal_nan=True), name
    # the sweep as its visits
    a.stage_spiral(-1.5)
    center = n // 2 - 1
    c.layer("groundpatch")[center, center] = 1.0
    c.layer("ground")[center, center] = np.float32(-1.5)
    visits = 0
    for i in range(center - 1, 0, -1):
        rp, sl = i, (center - i) * 2
        for side in range(2):
            for pos in range(rp, rp + sl):
                c.interpolate_cell(pos if side else rp, rp if side else pos)
                visits += 1
        R = rp + sl
        for side in range(2):
            for pos in range(R, R - sl - 1, -1):
                c.interpolate_cell(pos if side else R, R if side else pos)
                visits += 1
    assert visits == oracle.spiral_visit_count(n)
    for name in ("ground", "groundpatch"):
        assert np.array_equal(a.layer(name), c.layer(name), equal_nan=True), name
