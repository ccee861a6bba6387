"""The terrain sweep as the device runs it (ring-per-lane dataflow, groundgrid_amd/csrc/sweep_core.h), checked without a GPU.

The per-lane code of the gfx950 kernel is compiled for the host as well and executed by a lock-step emulation of its
wavefronts (gg_debug_emulate_ring_sweep): chains of 64 rings per wavefront, corner lanes, LDS hand-overs guarded by
progress counters.  The wavefronts are interleaved adversarially (seeded bursts: one wavefront runs as far as the
dataflow lets it while the others stand still) and layer loads can be resolved as late as their use (worst case for
write-after-read hazards on the in-place layer).  Whatever the interleaving, the result must equal the oracle's serial
sweep (src/GroundSegmentation.cpp:398-465) bit for bit, and the wavefronts must never deadlock."""
import ctypes as C

import numpy as np
import pytest

from groundgrid_amd import _lib, build
from oracle import oracle


@pytest.fixture(scope="module")
def lib():
    build.build()
    L = C.CDLL(_lib.LIB_PATH)
    L.gg_debug_emulate_ring_sweep.restype = C.c_int
    L.gg_debug_emulate_ring_sweep.argtypes = [C.c_int, C.c_double, C.c_float, C.c_void_p, C.c_float, C.c_double, C.c_uint, C.c_int, C.POINTER(C.c_long)]
    return L


def emulate(L, n, resolution, ground, conf, base_z, decrease, seed, late, min_dist_sq=12.0):
    gp2 = np.empty((n * n, 2), dtype=np.float32)
    gp2[:, 0] = ground.ravel(order="F")  # Eigen layers are column-major
    gp2[:, 1] = conf.ravel(order="F")
    stats = (C.c_long * 8)()
    rc = L.gg_debug_emulate_ring_sweep(n, resolution, min_dist_sq, gp2.ctypes.data, base_z, decrease, seed, int(late), stats)
    assert rc == 0, f"the wavefronts deadlocked ({rc})"
    return gp2[:, 0].reshape((n, n), order="F"), gp2[:, 1].reshape((n, n), order="F"), list(stats)


def random_state(n, seed):
    rng = np.random.default_rng(seed)
    ground = rng.normal(-1.7, 0.4, (n, n)).astype(np.float32)
    conf = rng.random((n, n)).astype(np.float32)
    conf[rng.random((n, n)) < 0.3] = 0.0          # cells without any estimate yet
    conf[rng.random((n, n)) < 0.05] = 1.0
    ground[rng.random((n, n)) < 0.01] = np.float32(37.5)
    return ground, conf


@pytest.mark.parametrize("length,resolution", [(4.0, 0.33), (5.0, 0.5), (10.0, 0.5), (33.0, 0.33), (43.0, 0.33), (61.0, 0.25), (120.0, 0.33),
                                               (150.0, 0.25), (240.0, 0.33)])
def test_ring_sweep_reproduces_the_serial_sweep(lib, length, resolution):
    ref = oracle.OracleMap(length, resolution)
    n = ref.layer("ground").shape[0]
    ground, conf = random_state(n, n)
    decrease = float(ref.cfg.occupied_cells_decrease_factor)
    ref.set_layer("ground", ground)
    ref.set_layer("groundpatch", conf)
    ref.stage_spiral(-1.73)
    seeds = (0, 1, 2, 3) if n <= 400 else (0, 5)
    for seed in seeds:
        for late in (False, True):
            g, w, stats = emulate(lib, n, ref.resolution, ground, conf, -1.73, decrease, seed, late)
            assert np.array_equal(g, ref.layer("ground")), (seed, late, np.argwhere(g != ref.layer("ground"))[:5].tolist())
            assert np.array_equal(w, ref.layer("groundpatch")), (seed, late)
    # cost model of the schedule: every visited cell is stored once (corners: the revisit only), loaded twice (own + outer line)
    visits = oracle.lib().ggo_spiral_visit_count(n)
    rings = n // 2 - 2
    assert stats[3] == visits - 2 * rings


@pytest.mark.parametrize("decrease", [1.1, 1.25, 5.0, 0.5, 7.3])
def test_decay_factors_and_decay_radius(lib, decrease):
    ref = oracle.OracleMap(33.0, 0.33)
    n = ref.layer("ground").shape[0]
    ground, conf = random_state(n, 77)
    ref.cfg.occupied_cells_decrease_factor = decrease
    ref.set_layer("ground", ground)
    ref.set_layer("groundpatch", conf)
    ref.stage_spiral(0.4)
    g, w, stats = emulate(lib, n, ref.resolution, ground, conf, 0.4, decrease, 9, True)
    assert np.array_equal(g, ref.layer("ground")) and np.array_equal(w, ref.layer("groundpatch"))
    # the decay radius as an integer threshold: 12 m^2 / (0.33 m)^2 = 110.19... -> cells with dx^2 + dy^2 >= 111 decay
    assert stats[7] == 111


def test_special_values_travel_unchanged(lib):
    """NaN / inf grounds and zero confidences through every hand-over path (lane to lane, LDS, memory)."""
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
    g, w, _ = emulate(lib, n, ref.resolution, ground, conf, -1.0, 5.0, 4, True)
    assert np.array_equal(g, ref.layer("ground"), equal_nan=True) and np.array_equal(w, ref.layer("groundpatch"), equal_nan=True)


@pytest.mark.parametrize("length,resolution,gpw", [(61.0, 0.25, 1), (120.0, 0.33, 1), (120.0, 0.33, 2), (150.0, 0.25, 2), (240.0, 0.33, 3), (240.0, 0.33, 1)])
def test_sweep_cut_into_several_work_groups(lib, monkeypatch, length, resolution, gpw):
    """sweep_core.h "Parts": a map with more ring groups than one work-group has wavefronts for is swept by several work-groups,
    each owning `gpw` consecutive groups of 64 rings; what crosses between them (boundary chains, corner values, two joins) is
    feed-forward and travels through a tagged exchange region that an importer wavefront republishes in the consumer's LDS.
    Same bar as the single work-group: bit-identical to the serial sweep under every interleaving, no deadlock."""
    monkeypatch.setenv("GG_SWEEP_GPW", str(gpw))
    ref = oracle.OracleMap(length, resolution)
    n = ref.layer("ground").shape[0]
    ground, conf = random_state(n, 3 * n + gpw)
    ref.set_layer("ground", ground)
    ref.set_layer("groundpatch", conf)
    ref.stage_spiral(-1.73)
    groups = (n // 2 - 2 + 63) // 64
    for seed in ((0, 1, 2, 3, 4) if n <= 400 else (0, 6)):
        for late in (False, True):
            g, w, stats = emulate(lib, n, ref.resolution, ground, conf, -1.73, 5.0, seed, late)
            assert np.array_equal(g, ref.layer("ground")), (seed, late, np.argwhere(g != ref.layer("ground"))[:5].tolist())
            assert np.array_equal(w, ref.layer("groundpatch")), (seed, late)
    parts = (groups + gpw - 1) // gpw
    assert parts >= 2 and stats[6] >= parts * 6 + (parts - 1)  # at least 4 chain + 2 corner wavefronts per part, an importer per hand-over


@pytest.mark.parametrize("length,resolution", [(22.0, 0.33), (61.0, 0.25), (120.0, 0.33), (240.0, 0.33)])
def test_split_steps_with_a_preparing_wavefront_per_side(lib, monkeypatch, length, resolution):
    """sweep_core.h "Split steps" (latency launches, one ring group per work-group): the layer half of every wave-step -- loads,
    confidence decay, products -- runs on a PREPARING wavefront a few steps ahead and reaches the chain wavefront through an LDS
    ring guarded by two counters.  The preparing wavefront loads cells up to PREP_DEPTH + PF steps before they are used: with
    `late` loads off that is the earliest a value can be read, with the adversarial schedule the ring runs full and empty."""
    monkeypatch.setenv("GG_SWEEP_GPW", "1")
    monkeypatch.setenv("GG_SWEEP_SPLIT", "1")
    ref = oracle.OracleMap(length, resolution)
    n = ref.layer("ground").shape[0]
    ground, conf = random_state(n, 5 * n + 1)
    ref.set_layer("ground", ground)
    ref.set_layer("groundpatch", conf)
    ref.stage_spiral(-1.73)
    for seed in ((0, 1, 2, 3, 4, 5) if n <= 400 else (0, 7)):
        for late in (False, True):
            g, w, stats = emulate(lib, n, ref.resolution, ground, conf, -1.73, 5.0, seed, late)
            assert np.array_equal(g, ref.layer("ground")), (seed, late, np.argwhere(g != ref.layer("ground"))[:5].tolist())
            assert np.array_equal(w, ref.layer("groundpatch")), (seed, late)


@pytest.mark.parametrize("n", [12, 30, 100, 130, 131, 258, 364, 727, 1000])
def test_one_compare_wait_test_is_the_inverse_of_the_closed_form_needs(lib, n):
    """k_sweep asks "may this half step run" with one scalar compare against the last step the cached counters cover; the closed forms
    (which ring's corner, which join, how much of the previous group's chain a step needs) run only where a counter is re-read."""
    lib.gg_debug_sweep_sync_selftest.restype = C.c_long
    lib.gg_debug_sweep_sync_selftest.argtypes = [C.c_int]
    assert lib.gg_debug_sweep_sync_selftest(n) == 0


# ---------------------------------------------------------------- FRESH maps (k4_sweep.hip run_chain<FRESH>, sweep_core.h step_a<.., FRESH>)

def emulate_fresh(L, n, resolution, ground, conf, patched, fresh_ground, base_z, decrease, seed, min_dist_sq=12.0):
    L.gg_debug_emulate_ring_sweep_fresh.restype = C.c_int
    L.gg_debug_emulate_ring_sweep_fresh.argtypes = [C.c_int, C.c_double, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_double, C.c_uint, C.POINTER(C.c_long)]
    gp2 = np.empty((n * n, 2), dtype=np.float32)
    gp2[:, 0] = ground.ravel(order="F")
    gp2[:, 1] = conf.ravel(order="F")
    mask = np.ascontiguousarray(patched.astype(np.uint8).ravel(order="F"))
    stats = (C.c_long * 8)()
    rc = L.gg_debug_emulate_ring_sweep_fresh(n, resolution, min_dist_sq, gp2.ctypes.data, mask.ctypes.data, fresh_ground, base_z, decrease, seed, stats)
    assert rc == 0, f"deadlock (-10) or a plan fault (-11): {rc}"
    return gp2[:, 0].reshape((n, n), order="F"), gp2[:, 1].reshape((n, n), order="F"), list(stats)


@pytest.mark.parametrize("length,resolution", [(4.0, 0.33), (10.0, 0.5), (22.0, 0.33), (23.0, 0.33), (43.0, 0.33), (61.0, 0.25), (120.0, 0.33), (150.0, 0.25)])
def test_fresh_map_sweep_reads_only_marked_cells(lib, length, resolution):
    """A FRESH map (gg_reset_maps wrote no cell): only the cells k_patch marked are in memory, every other cell holds the reset's pair by
    definition and the emulation POISONS its memory (NaN).  The sweep -- chains and corner lanes -- must take marked cells from memory and
    everything else from the one padding element, never the poison; the cells no sweep visits (ring >= c) come out real as well, with the
    reset's pair unless they are marked (an odd number of rows: the patch loops reach one line into that ring).  Result = the oracle's
    serial sweep of the layer as defined, bit for bit; even and odd sizes, one to four ring groups."""
    ref = oracle.OracleMap(length, resolution)
    n = ref.layer("ground").shape[0]
    rng = np.random.default_rng(1000 + n)
    fresh_ground = np.float32(rng.uniform(-0.5, 0.5))
    ground, conf = random_state(n, n + 1)
    # what k_patch may have written: rows and columns 2 .. n - 3, in clusters
    patched = np.zeros((n, n), dtype=bool)
    for _ in range(max(3, n // 6)):
        r0, c0 = rng.integers(2, n - 2, size=2)
        h, w = rng.integers(1, max(2, n // 4), size=2)
        patched[r0:min(r0 + h, n - 2), c0:min(c0 + w, n - 2)] = True
    patched &= rng.random((n, n)) < 0.8
    patched[:2, :] = patched[n - 2:, :] = False
    patched[:, :2] = patched[:, n - 2:] = False
    defined_g = np.where(patched, ground, fresh_ground).astype(np.float32)
    defined_w = np.where(patched, conf, np.float32(0.0000001)).astype(np.float32)
    decrease = float(ref.cfg.occupied_cells_decrease_factor)
    ref.set_layer("ground", defined_g)
    ref.set_layer("groundpatch", defined_w)
    ref.stage_spiral(-1.73)
    for seed in ((0, 1, 2, 3) if n <= 400 else (0, 5)):
        g, w, _ = emulate_fresh(lib, n, ref.resolution, ground, conf, patched, float(fresh_ground), -1.73, decrease, seed)
        assert not np.isnan(g).any() and not np.isnan(w).any(), (seed, "poison came through", np.argwhere(np.isnan(g))[:5].tolist())
        assert np.array_equal(g, ref.layer("ground")), (seed, np.argwhere(g != ref.layer("ground"))[:5].tolist())
        assert np.array_equal(w, ref.layer("groundpatch")), seed
