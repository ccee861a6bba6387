"""The pair sweep (groundgrid_amd/csrc/sweep_pair.h: the latency launches' terrain sweep) checked without a GPU.

A wavefront owns BOTH sides of a ring hand-over -- lanes 0..31 walk side A (or B), lanes 32..63 side D (or C) of the same 32 rings -- so the
joins are lane exchanges; what a visit needs of OLD cells comes from records a preparation pass wrote (the sweep never reads the layer).
The per-lane code of the gfx950 kernel runs here under a lock-step emulation with adversarial interleaving of the wavefronts (one or two
work-groups per cloud, any number of wavefronts per pair); whatever the interleaving, the result must equal the oracle's serial sweep
(src/GroundSegmentation.cpp:398-465) bit for bit, nothing may be read before it was published, and the wavefronts must never deadlock."""
import ctypes as C

import numpy as np
import pytest

from groundgrid_amd import _lib, build
from oracle import oracle


@pytest.fixture(scope="module")
def lib():
    build.build()
    L = C.CDLL(_lib.LIB_PATH)
    L.gg_debug_emulate_pair_sweep.restype = C.c_int
    L.gg_debug_emulate_pair_sweep.argtypes = [C.c_int, C.c_double, C.c_float, C.c_void_p, C.c_float, C.c_double, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_long)]
    return L


def emulate(L, n, resolution, ground, conf, base_z, decrease, seed, wgs=1, waves=0, min_dist_sq=12.0):
    gp2 = np.empty((n * n, 2), dtype=np.float32)
    gp2[:, 0] = ground.ravel(order="F")
    gp2[:, 1] = conf.ravel(order="F")
    stats = (C.c_long * 8)()
    rc = L.gg_debug_emulate_pair_sweep(n, resolution, min_dist_sq, gp2.ctypes.data, base_z, decrease, seed, wgs, waves, stats)
    assert rc == 0, f"deadlock (-10) or a plan / publication fault (-11): {rc}"
    return gp2[:, 0].reshape((n, n), order="F"), gp2[:, 1].reshape((n, n), order="F"), list(stats)


def random_state(n, seed):
    rng = np.random.default_rng(seed)
    ground = rng.normal(-1.7, 0.4, (n, n)).astype(np.float32)
    conf = rng.random((n, n)).astype(np.float32)
    conf[rng.random((n, n)) < 0.3] = 0.0
    conf[rng.random((n, n)) < 0.05] = 1.0
    ground[rng.random((n, n)) < 0.01] = np.float32(37.5)
    return ground, conf


def reference(length, resolution, seed, base_z=-1.73, decrease=None):
    ref = oracle.OracleMap(length, resolution)
    n = ref.layer("ground").shape[0]
    ground, conf = random_state(n, seed)
    if decrease is not None:
        ref.cfg.occupied_cells_decrease_factor = decrease
    ref.set_layer("ground", ground)
    ref.set_layer("groundpatch", conf)
    ref.stage_spiral(base_z)
    return ref, n, ground, conf


@pytest.mark.parametrize("length,resolution", [(4.0, 0.33), (5.0, 0.5), (10.0, 0.5), (22.0, 0.33), (23.0, 0.33), (33.0, 0.33), (43.0, 0.33), (61.0, 0.25),
                                               (120.0, 0.33), (150.0, 0.25), (240.0, 0.33)])
def test_pair_sweep_reproduces_the_serial_sweep(lib, length, resolution):
    ref, n, ground, conf = reference(length, resolution, 11 * int(length))
    decrease = float(ref.cfg.occupied_cells_decrease_factor)
    for seed in ((0, 1, 2, 3) if n <= 400 else (0, 5)):
        for wgs in (1, 2):
            g, w, stats = emulate(lib, n, ref.resolution, ground, conf, -1.73, decrease, seed, wgs)
            assert np.array_equal(g, ref.layer("ground")), (seed, wgs, np.argwhere(g != ref.layer("ground"))[:5].tolist())
            assert np.array_equal(w, ref.layer("groundpatch")), (seed, wgs, np.argwhere(w != ref.layer("groundpatch"))[:5].tolist())
    # every visit of the spiral is one record or one of the three corner visits per ring and corner
    rings = n // 2 - 2
    assert stats[2] == oracle.lib().ggo_spiral_visit_count(n) - 6 * rings


@pytest.mark.parametrize("waves", [1, 2, 3, 5])
def test_fewer_wavefronts_than_groups(lib, waves):
    """A wavefront that owns several 32-ring groups takes them in increasing order; the group outside waits for it (feed-forward only)."""
    ref, n, ground, conf = reference(120.0, 0.33, 91)
    for seed in (0, 7, 8):
        for wgs in (1, 2):
            g, w, stats = emulate(lib, n, ref.resolution, ground, conf, -1.73, 5.0, seed, wgs, waves)
            assert np.array_equal(g, ref.layer("ground")) and np.array_equal(w, ref.layer("groundpatch")), (seed, wgs)
    assert stats[7] == 6  # 180 rings in groups of 32


@pytest.mark.parametrize("decrease", [1.1, 1.25, 5.0, 0.5, 7.3])
def test_decay_factors(lib, decrease):
    ref, n, ground, conf = reference(33.0, 0.33, 77, base_z=0.4, decrease=decrease)
    g, w, _ = emulate(lib, n, ref.resolution, ground, conf, 0.4, decrease, 9, 2)
    assert np.array_equal(g, ref.layer("ground")) and np.array_equal(w, ref.layer("groundpatch"))


def test_special_values_travel_unchanged(lib):
    ref = oracle.OracleMap(61.0, 0.25)
    n = ref.layer("ground").shape[0]
    ground, conf = random_state(n, 5)
    rng = np.random.default_rng(6)
    ground[rng.random((n, n)) < 0.02] = np.nan
    ground[rng.random((n, n)) < 0.01] = np.inf
    conf[rng.random((n, n)) < 0.01] = np.float32(1e-30)
    ref.set_layer("ground", ground)
    ref.set_layer("groundpatch", conf)
    ref.stage_spiral(-1.0)
    for wgs in (1, 2):
        g, w, _ = emulate(lib, n, ref.resolution, ground, conf, -1.0, 5.0, 4, wgs)
        assert np.array_equal(g, ref.layer("ground"), equal_nan=True) and np.array_equal(w, ref.layer("groundpatch"), equal_nan=True)


# ---------------------------------------------------------------- the throughput launches' variant (sweep_pairb.h)

def emulate_batch(L, n, resolution, ground, conf, base_z, decrease, seed, late, waves=0, min_dist_sq=12.0):
    L.gg_debug_emulate_pair_sweep_batch.restype = C.c_int
    L.gg_debug_emulate_pair_sweep_batch.argtypes = [C.c_int, C.c_double, C.c_float, C.c_void_p, C.c_float, C.c_double, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_long)]
    gp2 = np.empty((n * n, 2), dtype=np.float32)
    gp2[:, 0] = ground.ravel(order="F")
    gp2[:, 1] = conf.ravel(order="F")
    stats = (C.c_long * 8)()
    rc = L.gg_debug_emulate_pair_sweep_batch(n, resolution, min_dist_sq, gp2.ctypes.data, base_z, decrease, seed, int(late), waves, stats)
    assert rc == 0, f"deadlock (-10) or a plan / publication fault (-11): {rc}"
    return gp2[:, 0].reshape((n, n), order="F"), gp2[:, 1].reshape((n, n), order="F"), list(stats)


@pytest.mark.parametrize("length,resolution", [(4.0, 0.33), (5.0, 0.5), (10.0, 0.5), (22.0, 0.33), (23.0, 0.33), (33.0, 0.33), (43.0, 0.33), (61.0, 0.25),
                                               (120.0, 0.33), (150.0, 0.25), (240.0, 0.33)])
def test_throughput_pair_sweep_reproduces_the_serial_sweep(lib, length, resolution):
    """sweep_pairb.h: the lanes of a pair wavefront load their own cells from the in-place layer, two steps ahead of their use -- or, in
    the emulation's `late` mode, at the very moment of use: if a cell could be rewritten before a visit that must still see its OLD
    value has read it, this finds it.  (confidence, product) pairs through the lane exchanges, LDS and the corner tables."""
    ref, n, ground, conf = reference(length, resolution, 13 * int(length))
    decrease = float(ref.cfg.occupied_cells_decrease_factor)
    for seed in ((0, 1, 2, 3) if n <= 400 else (0, 5)):
        for late in (False, True):
            g, w, stats = emulate_batch(lib, n, ref.resolution, ground, conf, -1.73, decrease, seed, late)
            assert np.array_equal(g, ref.layer("ground")), (seed, late, np.argwhere(g != ref.layer("ground"))[:5].tolist())
            assert np.array_equal(w, ref.layer("groundpatch")), (seed, late, np.argwhere(w != ref.layer("groundpatch"))[:5].tolist())
    visits = oracle.lib().ggo_spiral_visit_count(n)
    rings = n // 2 - 2
    assert stats[3] == visits - 2 * rings  # every visited cell is stored once (the corners: the revisit only)


@pytest.mark.parametrize("waves", [1, 2, 3])
def test_throughput_pair_sweep_with_fewer_wavefronts_than_groups(lib, waves):
    ref, n, ground, conf = reference(120.0, 0.33, 93)
    for seed in (0, 7, 8):
        for late in (False, True):
            g, w, _ = emulate_batch(lib, n, ref.resolution, ground, conf, -1.73, 5.0, seed, late, waves)
            assert np.array_equal(g, ref.layer("ground")) and np.array_equal(w, ref.layer("groundpatch")), (seed, late)


def test_throughput_pair_sweep_special_values_and_decay_factors(lib):
    ref = oracle.OracleMap(61.0, 0.25)
    n = ref.layer("ground").shape[0]
    ground, conf = random_state(n, 5)
    rng = np.random.default_rng(6)
    ground[rng.random((n, n)) < 0.02] = np.nan
    ground[rng.random((n, n)) < 0.01] = np.inf
    conf[rng.random((n, n)) < 0.01] = np.float32(1e-30)
    for decrease in (5.0, 1.1, 0.5):
        ref.cfg.occupied_cells_decrease_factor = decrease
        ref.set_layer("ground", ground)
        ref.set_layer("groundpatch", conf)
        ref.stage_spiral(-1.0)
        g, w, _ = emulate_batch(lib, n, ref.resolution, ground, conf, -1.0, decrease, 4, True)
        assert np.array_equal(g, ref.layer("ground"), equal_nan=True) and np.array_equal(w, ref.layer("groundpatch"), equal_nan=True), decrease
