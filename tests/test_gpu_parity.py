"""GPU parity tests (pytest -m gpu, on a real MI355X): the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs and against the committed vectors in tests/golden/.

Bar: integer results (class, cell, label, position in the returned cloud, counts) bit-exact; every float layer
bit-exact too (NaN == NaN), which is stricter than the 1e-4 m the north star allows for terrain height.
Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from groundgrid_amd import api, synth  # noqa: E402
from oracle import oracle  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ORIGIN0 = (0.0, 0.0, 0.0)


def nan_equal(a, b):
    return np.array_equal(a, b, equal_nan=True)


def assert_same_state(seg_map, ref, tag=""):
    for name in oracle.LAYERS:
        a, b = seg_map[name], ref.layer(name)
        if not nan_equal(a, b):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            raise AssertionError(f"{tag} layer {name}: {len(bad)} cells differ, first {bad[:3].tolist()} "
                                 f"gpu={[float(a[tuple(i)]) for i in bad[:3]]} ref={[float(b[tuple(i)]) for i in bad[:3]]}")


def run_pair(cloud, length=120.0, resolution=0.33, pos=(0.0, 0.0), origin=ORIGIN0, base_z=-1.73, frames=2, cfg_edit=None,
             odom_z=0.0):
    seg = api.GroundSegmentation().init(length, resolution, n_slots=1, max_points=max(len(cloud), 1))
    ref = oracle.OracleMap(length, resolution, pos=pos, odom_z=odom_z)
    seg.map(0).reset(odom_z=odom_z, pos=pos)
    if cfg_edit:
        c = seg.getConfig()
        cfg_edit(c)
        seg.setConfig(c)
        cfg_edit(ref.cfg)
    for f in range(frames):
        out, labels, index = seg.filter_cloud(cloud, origin, base_z, return_details=True)
        r = ref.filter_cloud(cloud, origin, base_z)
        cls, cell = seg.point_classes(len(cloud))
        assert np.array_equal(cls, r["cls"]), f"frame {f}: classes differ at {np.nonzero(cls != r['cls'])[0][:5]}"
        assert np.array_equal(cell, r["cell"]), f"frame {f}: cells differ"
        assert np.array_equal(labels, r["label"]), f"frame {f}: {int((labels != r['label']).sum())} labels differ"
        assert np.array_equal(index, r["index"]), f"frame {f}: returned-cloud order differs"
        assert out.tobytes() == r["out_points"].tobytes(), f"frame {f}: returned cloud differs"
        assert_same_state(seg.map(0), ref, f"frame {f}")
    seg.close()
    return r


# ---------------------------------------------------------------- committed vectors

@pytest.mark.parametrize("fname", sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz")))
def test_golden_vectors(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    cloud = np.frombuffer(g["cloud"].tobytes(), dtype=synth.POINT_DTYPE)
    seg = api.GroundSegmentation().init(float(g["length"]), float(g["resolution"]), n_slots=1, max_points=len(cloud))
    seg.map(0).reset(pos=tuple(g["pos"]))
    for f in range(int(g["frames"])):
        _, labels, index = seg.filter_cloud(cloud, tuple(g["origin"]), float(g["base_z"]), return_details=True)
        cls, _ = seg.point_classes(len(cloud))
        assert np.array_equal(labels, g[f"label_{f}"])
        assert np.array_equal(index, g[f"index_{f}"])
        assert np.array_equal(cls, g[f"cls_{f}"])
        for layer in ("ground", "groundpatch", "variance"):
            assert nan_equal(seg.map(0)[layer], g[f"{layer}_{f}"]), (f, layer)


# ---------------------------------------------------------------- oracle on the same seeded inputs

def test_expected_points_table_matches():
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=16)
    assert np.array_equal(seg.expected_points(), oracle.OracleMap(120.0, 0.33).expected_points())


def test_hdl64_full_size_three_frames():
    r = run_pair(synth.hdl64_cloud(), frames=3)  # BASELINE configs[1]: ~125 k points, 364 x 364
    assert (r["cls"] == oracle.OUTLIER).sum() > 0  # warm map: the line-of-sight test fired


def test_line_of_sight_walk_many_candidates():
    """k_classify runs the ray walk of :246-275 cooperatively (64 steps of one point per wavefront pass); it must
    decide exactly like the serial loop for short and long rays, rays leaving the map, and candidates in every lane."""
    base = synth.hdl64_cloud(seed=12, n_az=900)
    rng = np.random.default_rng(12)
    low = synth.clone_cloud(base)
    sel = rng.random(len(low)) < 0.2            # one point in five dives 0.3 .. 2.5 m under the surface
    low["z"][sel] -= rng.uniform(0.3, 2.5, sel.sum()).astype(np.float32)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(base))
    ref = oracle.OracleMap(120.0, 0.33)
    n_out = 0
    for f, (cloud, origin) in enumerate([(base, ORIGIN0), (low, ORIGIN0), (low, (7.5, -3.0, 0.4)), (low, (-80.0, 20.0, 1.0))]):
        _, labels, index = seg.filter_cloud(cloud, origin, -1.73, return_details=True)
        r = ref.filter_cloud(cloud, origin, -1.73)
        cls, _ = seg.point_classes(len(cloud))
        assert np.array_equal(cls, r["cls"]), f
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), f
        assert_same_state(seg.map(0), ref)
        n_out += int((r["cls"] == oracle.OUTLIER).sum())
    assert n_out > 1000


@pytest.mark.parametrize("length,resolution", [(4.0, 0.33), (10.0, 0.5), (33.0, 0.33), (61.0, 0.25), (150.0, 0.25), (240.0, 0.33)])
def test_geometries_from_tiny_to_wider_than_the_level_cap(length, resolution):
    """Everything derived from the geometry on the host -- sort tiles, R1 table, the layer's sheared element order, the
    sweep's ring groups and LDS hand-over tables (12 x 12 cells up to 727 x 727: 6 ring groups on 3 wavefronts per side) --
    must reproduce the CPU path; the cloud is scaled to cover the map."""
    base = synth.hdl64_cloud(seed=19, n_az=500)
    cloud = synth.clone_cloud(base)
    k = np.float32(length / 120.0)
    cloud["x"] *= k
    cloud["y"] *= k
    run_pair(cloud, length=length, resolution=resolution, frames=3)


def test_hdl64_firing_order():
    run_pair(synth.hdl64_cloud(seed=5, order="azimuth"), frames=2)  # every consecutive point in another cell/tile


def test_unstructured_cloud_moved_map_and_origin():
    run_pair(synth.random_cloud(60000, seed=9), pos=(4.29, -2.64), origin=(4.0, -2.5, 0.1), base_z=-1.6, frames=3)


def test_dense_single_cells_and_ties():
    # thousands of points in a handful of cells: long ordered Welford chains, LDS staging over several chunks
    rng = np.random.default_rng(4)
    n = 30000
    xy = rng.choice(np.array([5.0, 5.2, 5.4, 7.7]), size=(n, 2)) + rng.uniform(0, 0.05, size=(n, 2))
    z = rng.normal(-1.7, 0.05, size=n)
    run_pair(synth.make_cloud(np.column_stack([xy, z]), ring=rng.integers(0, 64, n)), frames=2)


def test_config_variations():
    def edit(c):
        c.max_ring = 40            # rings 41..63 ignored
        c.point_count_cell_variance_threshold = 3
        c.outlier_tolerance = 0.05
        c.patch_size_change_distance = 10.0
        c.occupied_cells_decrease_factor = 3.0
        c.min_outlier_detection_ground_confidence = 0.8
    r = run_pair(synth.hdl64_cloud(seed=21, n_az=700), frames=3, cfg_edit=edit)
    assert (r["cls"] == oracle.IGNORED).sum() > 0


@pytest.mark.parametrize("factor", [1.1, 1.25, 0.5, 7.3, 1e6])
def test_confidence_decay_factor_fast_and_exact_branches(factor):
    """k_sweep replaces the f64 divide of :463-464 by a guarded multiply for factors >= 1.25 (tests/test_oracle_cpu.py
    proves the guard); smaller factors take the divide.  Both must reproduce the CPU path over several frames."""
    def edit(c):
        c.occupied_cells_decrease_factor = factor
    run_pair(synth.hdl64_cloud(seed=33, n_az=500), frames=4, cfg_edit=edit)


def test_edge_cases():
    pts = np.array([[5, 5, -1], [1, 1, -1], [5, 5, -1], [500, 0, -1], [np.nan, 0, -1], [5, 5, np.nan], [-59.9, -59.9, -1],
                    [np.inf, 1, 0], [59.99, 59.99, 0.5], [0, 0, 3], [3, -59.5, -1.6], [5, 5, -np.inf], [-1e30, 2, 0]],
                   dtype=np.float32)
    cloud = synth.make_cloud(pts, ring=[0, 0, 2000, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5])
    run_pair(cloud, frames=2)
    # a SIGNALLING NaN height between two ordinary returns of one cell (np.nan is a quiet one): std::max(mx, z) of :307 must leave the
    # cell's maximum alone (k2_reduce.hip quiet)
    snan = synth.make_cloud(np.array([[5, 5, -1.2], [5.01, 5.01, 0.0], [5.02, 5.0, -1.1], [7, 7, 0.0], [7.01, 7.0, -1.0]], dtype=np.float32))
    z = snan["z"].view(np.uint32)
    z[1] = 0x7FA00000
    z[3] = 0xFF800001
    assert np.isnan(snan["z"][1]) and np.isnan(snan["z"][3])
    run_pair(snan, frames=2)
    run_pair(synth.empty_cloud(0), frames=2)
    run_pair(synth.make_cloud(np.array([[900.0, 900.0, 0.0]], dtype=np.float32)), frames=1)  # everything outside


def test_initial_ground_height_nonzero():
    run_pair(synth.hdl64_cloud(seed=3, n_az=400), frames=2, odom_z=-1.5)


def test_config4_geometry_dense_cloud():
    # BASELINE configs[3] geometry: 200 m @ 0.2 m -> 1000 x 1000 cells; 128 beams x 2048 azimuths (~260 k points)
    cloud = synth.os128_cloud(seed=1, n_az=2048)
    seg = api.GroundSegmentation().init(200.0, 0.2, n_slots=1, max_points=len(cloud))
    assert (seg.rows, seg.cols) == (1000, 1000)
    seg.close()
    run_pair(cloud, length=200.0, resolution=0.2, frames=2)


# ---------------------------------------------------------------- batched, device-resident entry point

def _batch_inputs(fmt, clouds, stride):
    import torch

    B = len(clouds)
    if fmt == 16:
        host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
        for b, c in enumerate(clouds):
            host[b, : len(c)] = api.pack16(c)
        raw = host.view(np.uint8).reshape(B, stride, 16)
    else:
        raw = np.zeros((B, stride, 32), dtype=np.uint8)
        for b, c in enumerate(clouds):
            raw[b, : len(c)] = np.frombuffer(c.tobytes(), dtype=np.uint8).reshape(-1, 32)
    return torch.from_numpy(raw).cuda()


@pytest.mark.parametrize("fmt", [16, 32])
def test_batched_independent_maps(fmt):
    import torch

    clouds = [synth.hdl64_cloud(seed=100 + k, n_az=300 + 37 * k) for k in range(5)] + [synth.empty_cloud(0)]
    B, stride = len(clouds), max(len(c) for c in clouds) + 3
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B + 1, max_points=stride)
    refs = [oracle.OracleMap(120.0, 0.33) for _ in clouds]
    pts = _batch_inputs(fmt, clouds, stride)
    origins = np.array([[0.1 * b, -0.05 * b, 0.02 * b] for b in range(B)], dtype=np.float32)
    base_z = np.array([-1.73 + 0.01 * b for b in range(B)])
    out = None
    for frame in range(2):
        out = seg.filter_batch(pts, [len(c) for c in clouds], origins, base_z, first_slot=1, out=out, want_clouds=(fmt == 32))
        torch.cuda.synchronize()
        labels, index, counts = out.labels.cpu().numpy(), out.out_index.cpu().numpy(), out.counts.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, tuple(origins[b]), float(base_z[b]))
            n = len(c)
            assert np.array_equal(labels[b, :n], r["label"]), (frame, b)
            assert np.array_equal(index[b, :n], r["index"]), (frame, b)
            assert counts[b, 0] == len(r["out_points"])
            assert counts[b, 3] == (r["cls"] == oracle.OUTLIER).sum()
            assert_same_state(seg.map(b + 1), refs[b], f"frame {frame} cloud {b}")
            if fmt == 32:
                got = out.out_clouds[b, : counts[b, 0]].cpu().numpy().tobytes()
                assert got == r["out_points"].tobytes(), (frame, b)
    # slot 0 was never touched by the batch
    assert (seg.map(0)["ground"] == 0).all()


def _mixed_small_clouds(count, seed0):
    """Distinct small clouds (6-7 k points; every 16th ~36 k so that dense K2 tiles and several 2048-point chunks occur)."""
    return [synth.hdl64_cloud(seed=seed0 + k, n_az=(600 if k % 16 == 5 else 100 + (k % 7) * 3)) for k in range(count)]


def _check_batch_against_oracle(seg, clouds, pts, origins, base_z, frames, layer_slots, first_slot=0, tag=""):
    """filter_batch `frames` times; labels / emission index / counts of EVERY cloud and all 11 layers of `layer_slots`
    against one oracle map per cloud."""
    import torch

    refs = [oracle.OracleMap(120.0, 0.33) for _ in clouds]
    out = None
    for frame in range(frames):
        out = seg.filter_batch(pts, [len(c) for c in clouds], origins, base_z, first_slot=first_slot, out=out)
        torch.cuda.synchronize()
        labels, index, counts = out.labels.cpu().numpy(), out.out_index.cpu().numpy(), out.counts.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, tuple(origins[b]), float(base_z[b]))
            n = len(c)
            assert np.array_equal(labels[b, :n], r["label"]), (tag, frame, b)
            assert np.array_equal(index[b, :n], r["index"]), (tag, frame, b)
            assert counts[b, 0] == len(r["out_points"]), (tag, frame, b)
            assert counts[b, 3] == (r["cls"] == oracle.OUTLIER).sum(), (tag, frame, b)
            if b in layer_slots:
                assert_same_state(seg.map(first_slot + b), refs[b], f"{tag} frame {frame} cloud {b}")


def test_benchmark_launch_geometry_288_slots():
    """The launch geometry bench.py runs: a context with >= 128 slots (2048-point wave chunks in K1 / scan /
    scatter / K5), a launch of > 256 clouds (k_sweep with make_params' throughput setting: 2 chain wavefronts per side, one
    wavefront walking two ring groups) and k_reduce at its floor of 64 work-groups per cloud -- all three at once, 288 DISTINCT
    clouds on 288 maps, two frames (the second on warm maps: outliers, liveness masks), every cloud's labels / order / counts
    and all 11 layers of 24 sampled slots against the oracle."""
    B = 288
    clouds = _mixed_small_clouds(B, 3000)
    clouds[7] = synth.empty_cloud(0)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
    assert seg.debug_set_tuning("pw", 0) == 2048
    pts = _batch_inputs(16, clouds, stride)
    origins = np.array([[0.05 * (b % 9), -0.03 * (b % 5), 0.01 * (b % 3)] for b in range(B)], dtype=np.float32)
    base_z = np.array([-1.73 + 0.002 * (b % 11) for b in range(B)])
    sampled = set(range(0, B, 13)) | {5, 7, B - 1}
    _check_batch_against_oracle(seg, clouds, pts, origins, base_z, 2, sampled, tag="288 slots")
    seg.close()


@pytest.mark.parametrize("knob", ["pw2048", "sweep_waves1", "sweep_waves2", "sweep_waves3", "k2_per_cloud64", "k2_dense_share4", "all_big_batch"])
def test_each_launch_geometry_switch_forced_at_small_batch(knob, monkeypatch):
    """The same three switches one at a time (and together) on a 6-cloud batch, full-size clouds included, so that a failure
    names the switch: GG_PW=2048 at gg_create; sweep wavefronts per side 1 / 2 / 3; k_reduce with 64 work-groups per cloud and
    another dense / light split."""
    if knob in ("pw2048", "all_big_batch"):
        monkeypatch.setenv("GG_PW", "2048")
    clouds = [synth.hdl64_cloud(seed=410 + k, n_az=n) for k, n in enumerate([2083, 700, 150, 1200, 90])] + [synth.empty_cloud(0)]
    B, stride = len(clouds), (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
    if knob in ("pw2048", "all_big_batch"):
        assert seg.debug_set_tuning("pw", 0) == 2048
    if knob.startswith("sweep_waves"):
        seg.debug_set_tuning("sweep_waves", int(knob[-1]))
    if knob in ("k2_per_cloud64", "all_big_batch"):
        seg.debug_set_tuning("k2_per_cloud", 64)
    if knob == "k2_dense_share4":
        seg.debug_set_tuning("k2_dense_share", 4)
    if knob == "all_big_batch":
        seg.debug_set_tuning("sweep_waves", 2)
    pts = _batch_inputs(16, clouds, stride)
    origins = np.zeros((B, 3), dtype=np.float32)
    base_z = np.full(B, -1.73)
    _check_batch_against_oracle(seg, clouds, pts, origins, base_z, 3, set(range(B)), tag=knob)
    seg.close()


@pytest.mark.parametrize("shape", [1, 2, 3])
@pytest.mark.parametrize("pw,n_az,geometry", [(8192, 100, (120.0, 0.33)), (2048, 60, (120.0, 0.33)), (2048, 2083, (120.0, 0.33)), (64, 64, (120.0, 0.33)),
                                               (256, 900, (61.0, 0.25)), (0, 700, (200.0, 0.2)), (1024, 500, (200.0, 0.2))])
def test_front_end_in_one_two_and_three_launches(shape, pw, n_az, geometry, monkeypatch):
    """The front end (classify + stable tile sort) as three launches, with the scan inside k_classify (the last work-group of a
    cloud to finish scans it) and as ONE launch (ticketed work-groups wait for their cloud's scan and scatter their own chunks;
    k1_classify.hip FRONT_*), forced per context, with chunk sizes that give the largest cloud 1, 2 and ~60 chunks (and 15
    work-groups of 64-point chunks), on the 16-bit packed tile counters of maps with more than 1024 tiles (200 m / 0.2 m), and
    with clouds of very different sizes plus an empty one in the same launch (work-groups without chunks, a cloud nobody but
    its first work-group scans).  Three frames, every cloud's labels / order / counts and all 11 layers against the oracle."""
    if pw:
        monkeypatch.setenv("GG_PW", str(pw))
    length, resolution = geometry
    k = np.float32(length / 120.0)
    clouds = []
    for j, frac in enumerate([1.0, 0.31, 0.07, 0.55]):
        c = synth.clone_cloud(synth.hdl64_cloud(seed=880 + j, n_az=max(8, int(n_az * frac))))
        c["x"] *= k
        c["y"] *= k
        clouds.append(c)
    clouds.insert(2, synth.empty_cloud(0))
    B, stride = len(clouds), (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=B, max_points=stride)
    if pw:
        assert seg.debug_set_tuning("pw", 0) == pw
    seg.debug_set_tuning("front", shape)
    pts = _batch_inputs(16, clouds, stride)
    import torch

    refs = [oracle.OracleMap(length, resolution) for _ in clouds]
    out = None
    for frame in range(3):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73), out=out)
        torch.cuda.synchronize()
        labels, index, counts = out.labels.cpu().numpy(), out.out_index.cpu().numpy(), out.counts.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, ORIGIN0, -1.73)
            n = len(c)
            assert np.array_equal(labels[b, :n], r["label"]), (frame, b)
            assert np.array_equal(index[b, :n], r["index"]), (frame, b)
            assert counts[b, 0] == len(r["out_points"]) and counts[b, 3] == (r["cls"] == oracle.OUTLIER).sum(), (frame, b)
            assert_same_state(seg.map(b), refs[b], f"shape {shape} frame {frame} cloud {b}")
    seg.close()


@pytest.mark.parametrize("length,resolution,pw,parts", [(120.0, 0.33, 2048, 5), (120.0, 0.33, 64, 16), (200.0, 0.2, 0, 0), (200.0, 0.2, 1024, 3),
                                                         (200.0, 0.2, 256, 16), (280.0, 0.2, 0, 0), (280.0, 0.2, 512, 2), (280.0, 0.2, 2048, 13)])
def test_tile_scan_cut_into_several_work_groups(length, resolution, pw, parts, monkeypatch):
    """k_scan with several work-groups per cloud (sort_core.h scan_cloud "PARTS": a part sums its tile groups' columns, publishes
    its sums in one 64-bit word, waits for the parts before it): forced part counts 2 .. 16 (0 = the launcher's own choice for
    maps with thousands of tiles and few clouds) on 529, 3969 and 7744 tiles, part ranges that are ragged or shorter than a
    wavefront, 1 .. ~270 chunk rows, clouds of very different sizes and an empty one in the same launch.  Two frames, labels /
    order / counts and all 11 layers against the oracle."""
    if pw:
        monkeypatch.setenv("GG_PW", str(pw))
    k = np.float32(length / 120.0)
    clouds = []
    for j, frac in enumerate([1.0, 0.2, 0.6]):
        c = synth.clone_cloud(synth.hdl64_cloud(seed=930 + j, n_az=max(8, int(280 * frac))))
        c["x"] *= k
        c["y"] *= k
        clouds.append(c)
    clouds.insert(1, synth.empty_cloud(0))
    B, stride = len(clouds), (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=B, max_points=stride)
    seg.debug_set_tuning("scan_parts", parts)
    pts = _batch_inputs(16, clouds, stride)
    import torch

    refs = [oracle.OracleMap(length, resolution) for _ in clouds]
    out = None
    for frame in range(2):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73), out=out)
        torch.cuda.synchronize()
        seg.synchronize()
        labels, index, counts = out.labels.cpu().numpy(), out.out_index.cpu().numpy(), out.counts.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, ORIGIN0, -1.73)
            n = len(c)
            assert np.array_equal(labels[b, :n], r["label"]), (frame, b)
            assert np.array_equal(index[b, :n], r["index"]), (frame, b)
            assert counts[b, 0] == len(r["out_points"]) and counts[b, 3] == (r["cls"] == oracle.OUTLIER).sum(), (frame, b)
            assert_same_state(seg.map(b), refs[b], f"{parts} parts, frame {frame}, cloud {b}")
    seg.close()


def test_scan_part_that_never_publishes_is_an_error_not_a_hang():
    """The parts of a cloud's scan wait for the sums of the parts before them -- bounded: a part whose predecessor never
    publishes (debug knob) gives up after about half a second, the kernels run to their end and the next synchronising call
    reports GG_ERR_HIP with the reason; afterwards the context works again."""
    import torch

    clouds = _rotated_clouds(2, 120.0)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=2, max_points=stride)
    seg.debug_set_tuning("scan_parts", 4)
    seg.debug_set_tuning("scan_fault", 1)
    seg.debug_set_tuning("scan_poll_cap", 20000)  # (the default bound is several seconds)
    pts = _batch_inputs(16, clouds, stride)
    n, org, bz = [len(c) for c in clouds], np.zeros((2, 3), np.float32), np.full(2, -1.73)
    seg.filter_batch(pts, n, org, bz)
    torch.cuda.synchronize()  # (returns: the kernels ended)
    with pytest.raises(api.GroundGridError, match="k_scan: the sums of an earlier part"):
        seg.synchronize()
    seg.debug_set_tuning("scan_fault", 0)
    seg.debug_set_tuning("scan_poll_cap", 0)
    assert seg._L.gg_device_error(seg._ctx, 0) == 0  # reported once and cleared: the context works again by itself
    seg.reset_maps(0, 2)
    refs = [oracle.OracleMap(120.0, 0.33) for _ in clouds]
    out = seg.filter_batch(pts, n, org, bz)
    torch.cuda.synchronize()
    seg.synchronize()
    labels = out.labels.cpu().numpy()
    for b, c in enumerate(clouds):
        r = refs[b].filter_cloud(c, ORIGIN0, -1.73)
        assert np.array_equal(labels[b, : len(c)], r["label"]), b
        assert_same_state(seg.map(b), refs[b], f"after the fault, cloud {b}")
    seg.close()


def test_batch_with_a_slot_permutation():
    """gg_batch.slots (ABI v3): cloud b meets map slots[b]; the maps keep their own histories under changing permutations, and
    duplicate / out-of-range entries are rejected."""
    import torch

    clouds = [synth.hdl64_cloud(seed=520 + k, n_az=200 + 31 * k) for k in range(5)]
    B, stride = len(clouds), (max(len(c) for c in clouds) + 63) // 64 * 64
    n_slots = 9
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=n_slots, max_points=stride)
    refs = [oracle.OracleMap(120.0, 0.33) for _ in range(n_slots)]
    pts = _batch_inputs(16, clouds, stride)
    out = None
    for frame, slots in enumerate([[8, 0, 3, 5, 2], [0, 8, 2, 3, 1], [4, 3, 2, 1, 0]]):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73), out=out, slots=slots)
        torch.cuda.synchronize()
        labels, index = out.labels.cpu().numpy(), out.out_index.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[slots[b]].filter_cloud(c, ORIGIN0, -1.73)
            assert np.array_equal(labels[b, : len(c)], r["label"]) and np.array_equal(index[b, : len(c)], r["index"]), (frame, b)
        for s_ in range(n_slots):
            assert_same_state(seg.map(s_), refs[s_], f"frame {frame} slot {s_}")
    for bad in ([0, 1, 2, 3, 3], [0, 1, 2, 3, 9], [0, 1, 2, 3, -1]):
        with pytest.raises(api.GroundGridError):
            seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73), slots=bad)
    seg.close()


@pytest.mark.parametrize("length,resolution,gpw,batch,split", [(120.0, 0.33, 1, 3, 0), (120.0, 0.33, 2, 11, 0), (120.0, 0.33, 1, 17, 2), (240.0, 0.33, 0, 2, 0),
                                                               (240.0, 0.33, 2, 9, 0), (61.0, 0.25, 1, 8, 0), (120.0, 0.33, 1, 9, 1), (240.0, 0.33, 1, 3, 2),
                                                               (200.0, 0.2, 1, 1, 1), (22.0, 0.33, 1, 2, 1)])
def test_sweep_cut_into_several_work_groups(length, resolution, gpw, batch, split):
    """k_sweep as several cooperating work-groups per cloud (sweep_core.h "Parts": `gpw` ring groups of 64 each, values crossing
    through the tagged exchange region + importer wavefront): forced at the bench grid (3 groups -> 3 or 2 parts) and at a
    727-cell map (6 groups: the default cut, 3 + 3), with batch sizes on both sides of the 8-cloud bundles the work-group
    index is laid out in.  `split`: the "split steps" of launches with one ring group per work-group (a preparing wavefront per
    side does the layer half of every step; sweep_core.h) forced on (1) / off (2) / left to the launcher (0: on for these small
    launches whenever gpw == 1).  Two frames, every layer against the oracle."""
    base = synth.hdl64_cloud(seed=61, n_az=400)
    k = np.float32(length / 120.0)
    clouds = []
    for b in range(batch):
        c = synth.clone_cloud(base)
        ang = np.float32(0.31 * b)
        c["x"] = ((np.cos(ang) * base["x"] - np.sin(ang) * base["y"]) * k).astype(np.float32)
        c["y"] = ((np.sin(ang) * base["x"] + np.cos(ang) * base["y"]) * k).astype(np.float32)
        clouds.append(c)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
    if gpw:
        seg.debug_set_tuning("sweep_gpw", gpw)
    if split:
        seg.debug_set_tuning("sweep_split", split)
    import torch

    refs = [oracle.OracleMap(length, resolution) for _ in clouds]
    pts = _batch_inputs(16, clouds, stride)
    out = None
    for frame in range(2):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((batch, 3), np.float32), np.full(batch, -1.73), out=out)
        torch.cuda.synchronize()
        labels = out.labels.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, ORIGIN0, -1.73)
            assert np.array_equal(labels[b, : len(c)], r["label"]), (frame, b)
            if b in (0, 1, batch // 2, batch - 1):
                assert_same_state(seg.map(b), refs[b], f"frame {frame} cloud {b}")
    seg.close()


def _rotated_clouds(count, length, seed=61, n_az=400):
    base = synth.hdl64_cloud(seed=seed, n_az=n_az)
    k = np.float32(length / 120.0)
    clouds = []
    for b in range(count):
        c = synth.clone_cloud(base)
        ang = np.float32(0.31 * b)
        c["x"] = ((np.cos(ang) * base["x"] - np.sin(ang) * base["y"]) * k).astype(np.float32)
        c["y"] = ((np.sin(ang) * base["x"] + np.cos(ang) * base["y"]) * k).astype(np.float32)
        clouds.append(c)
    return clouds


@pytest.mark.parametrize("length,resolution,batch,split", [(120.0, 0.33, 3, 0), (120.0, 0.33, 11, 2), (240.0, 0.33, 5, 0)])
def test_sweep_parts_started_in_reverse_order_still_match(length, resolution, batch, split):
    """The work-groups that sweep one cloud take a ticket when they start running, and the ticket names (cloud, part): a part's
    producer has always started before it, whatever order the dispatcher starts work-groups in (MI355X_MICROARCH.md Contract
    [G]).  The debug knob hands the tickets out in REVERSE -- every consumer starts before its producer, the worst order an
    all-resident launch can see -- and the results must not change: labels of every cloud and all layers against the oracle,
    two frames."""
    import torch

    clouds = _rotated_clouds(batch, length)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
    seg.debug_set_tuning("sweep_gpw", 1)
    if split:
        seg.debug_set_tuning("sweep_split", split)
    seg.debug_set_tuning("sweep_fault", 1)
    refs = [oracle.OracleMap(length, resolution) for _ in clouds]
    pts = _batch_inputs(16, clouds, stride)
    out = None
    for frame in range(2):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((batch, 3), np.float32), np.full(batch, -1.73), out=out)
        torch.cuda.synchronize()
        seg.synchronize()
        labels = out.labels.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, ORIGIN0, -1.73)
            assert np.array_equal(labels[b, : len(c)], r["label"]), (frame, b)
            assert_same_state(seg.map(b), refs[b], f"frame {frame} cloud {b}")
    seg.close()


def test_sweep_hand_over_that_never_arrives_is_an_error_not_a_hang():
    """A part of a multi-work-group sweep whose producer never delivers (here: the exporter withholds the joins of its last
    ring) must not spin forever: the importer's wait is bounded, the kernel runs to its end, and the next call that
    synchronises reports GG_ERR_HIP with the reason.  Afterwards the context works again (error cleared, maps reset)."""
    import torch

    clouds = _rotated_clouds(2, 120.0)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=2, max_points=stride)
    seg.debug_set_tuning("sweep_gpw", 1)
    seg.debug_set_tuning("sweep_fault", 2)
    seg.debug_set_tuning("sweep_poll_cap", 3000)
    pts = _batch_inputs(16, clouds, stride)
    n, org, bz = [len(c) for c in clouds], np.zeros((2, 3), np.float32), np.full(2, -1.73)
    seg.filter_batch(pts, n, org, bz)
    torch.cuda.synchronize()  # (returns: the kernels ended)
    with pytest.raises(api.GroundGridError, match="k_sweep: a hand-over"):
        seg.synchronize()
    seg.debug_set_tuning("sweep_fault", 0)
    seg.debug_set_tuning("sweep_poll_cap", 0)
    assert seg._L.gg_device_error(seg._ctx, 0) == 0  # reported once and cleared
    seg.reset_maps(0, 2)
    refs = [oracle.OracleMap(120.0, 0.33) for _ in clouds]
    out = seg.filter_batch(pts, n, org, bz)
    torch.cuda.synchronize()
    seg.synchronize()
    labels = out.labels.cpu().numpy()
    for b, c in enumerate(clouds):
        r = refs[b].filter_cloud(c, ORIGIN0, -1.73)
        assert np.array_equal(labels[b, : len(c)], r["label"]), b
        assert_same_state(seg.map(b), refs[b], f"after the fault, cloud {b}")
    seg.close()


def test_two_bit_label_masks_from_the_label_kernel():
    """gg_batch.d_label_masks: what a multi-GPU caller all-gathers (groundgrid_amd/dist.py) -- the labels, 2 bits per point."""
    import torch
    from groundgrid_amd.dist import pack_label_masks, unpack_label_masks

    clouds = [synth.hdl64_cloud(seed=200 + k, n_az=250 + 41 * k) for k in range(4)] + [synth.empty_cloud(0)]
    B, stride = len(clouds), (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
    pts = _batch_inputs(16, clouds, stride)
    out = None
    for _ in range(2):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73), out=out, want_masks=True)
    torch.cuda.synchronize()
    labels = out.labels.cpu()
    for b, c in enumerate(clouds):
        n = len(c)
        nb = (n + 3) // 4
        want = torch.zeros(stride, dtype=torch.uint8)
        want[:n] = labels[b, :n]
        assert torch.equal(out.label_masks[b, :nb].cpu(), pack_label_masks(want[None, :])[0, :nb]), b
        assert torch.equal(unpack_label_masks(out.label_masks[b : b + 1].cpu(), stride)[0, :n], labels[b, :n]), b
    with pytest.raises(api.GroundGridError):  # a stride that is not a multiple of 4 cannot carry masks
        bad = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=1002)
        bad.filter_batch(_batch_inputs(16, [synth.empty_cloud(0)], 1002), [0], np.zeros((1, 3), np.float32), np.full(1, -1.73), want_masks=True)


def test_minimal_layers_flag_keeps_labels_and_terrain():
    cloud = synth.hdl64_cloud(seed=8, n_az=600)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    seg.set_flags(minimal_layers=True)
    ref = oracle.OracleMap(120.0, 0.33)
    for _ in range(2):
        _, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
        r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"])
        for name in ("ground", "groundpatch", "points", "variance", "m2", "minGroundHeight", "pointsRaw", "meanVariance"):
            assert nan_equal(seg.map(0)[name], ref.layer(name)), name
    # back to all layers: the three that were not maintained are rewritten everywhere by the next cloud
    seg.set_flags(minimal_layers=False)
    moved = synth.hdl64_cloud(seed=9, n_az=600)
    _, labels, _ = seg.filter_cloud(moved, ORIGIN0, -1.73, return_details=True)
    r = ref.filter_cloud(moved, ORIGIN0, -1.73)
    assert np.array_equal(labels, r["label"])
    assert_same_state(seg.map(0), ref, "after leaving minimal layers")


def test_minimal_layers_materialise_the_other_three_on_demand():
    """GG_FLAG_MINIMAL_LAYERS (lazily materialised layers): k_reduce maintains only the six per-call layers the
    path reads; maxGroundHeight / groundCandidates / planeDist (src/GroundSegmentation.cpp:296,303,307) are computed when somebody
    asks, from the tile-sorted records the call left behind.  All 11 layers against the oracle through gg_get_layer, gg_get_layers and
    the 8-bit image; light and dense tiles (sensor cloud, thousands of points in four cells, NaN heights); several slots of one batch,
    asked in another order than they ran; a host write in between; a second read costs nothing and changes nothing."""
    rng = np.random.default_rng(14)
    n = 20000
    xy = rng.choice(np.array([5.0, 5.2, 5.4, 7.7]), size=(n, 2)) + rng.uniform(0, 0.05, size=(n, 2))
    dense = synth.make_cloud(np.column_stack([xy, rng.normal(-1.7, 0.05, size=n)]), ring=rng.integers(0, 64, n))
    dense["z"][::977] = np.float32("nan")
    clouds = [synth.hdl64_cloud(seed=8, n_az=600), dense, synth.random_cloud(3000, seed=3, extent=50.0), synth.empty_cloud(0)]
    for cloud in clouds:
        seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=max(len(cloud), 1))
        seg.set_flags(minimal_layers=True)
        ref = oracle.OracleMap(120.0, 0.33)
        for f in range(2):
            _, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
            r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
            assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"])
            if f == 0:
                assert_same_state(seg.map(0), ref, "one layer at a time")  # (gg_get_layer, in the layers' own order)
            else:
                got = seg.map(0).layers()  # (gg_get_layers: one call)
                for name in oracle.LAYERS:
                    assert nan_equal(got[name], ref.layer(name)), name
            assert_same_state(seg.map(0), ref, "second read")
        img, lo, hi = seg.map(0).image_u8("maxGroundHeight")
        want = ref.layer("maxGroundHeight")
        assert lo == np.nanmin(want) and hi == np.nanmax(want)
        seg.close()
    # a batch of four slots; the layers of slot 2 are asked first, slot 0 only after another cloud went through slot 1
    B = 4
    batch = [synth.hdl64_cloud(seed=40 + b, n_az=300 + 50 * b) for b in range(B)]
    stride = (max(len(c) for c in batch) + 3) // 4 * 4
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
    seg.set_flags(minimal_layers=True)
    refs = [oracle.OracleMap(120.0, 0.33) for _ in range(B)]
    for b in range(B):
        seg.map(b).reset()
    seg.filter_batch(_batch_inputs(16, batch, stride), [len(c) for c in batch], np.zeros((B, 3), np.float32), np.full(B, -1.73))
    for b in range(B):
        refs[b].filter_cloud(batch[b], ORIGIN0, -1.73)
    assert_same_state(seg.map(2), refs[2], "slot 2")
    seg.filter_cloud(batch[3], ORIGIN0, -1.73, map=seg.map(1))
    refs[1].filter_cloud(batch[3], ORIGIN0, -1.73)
    for b in (0, 1, 3, 2):
        assert_same_state(seg.map(b), refs[b], f"slot {b}")
    # a host write of a maintained layer densifies all nine: the three must have been computed before
    mine = np.full(refs[0].layer("m2").shape, 2.5, dtype=np.float32)
    seg.filter_cloud(batch[0], ORIGIN0, -1.73, map=seg.map(0))
    refs[0].filter_cloud(batch[0], ORIGIN0, -1.73)
    seg.map(0).set("m2", mine)
    for name in oracle.LAYERS:
        assert nan_equal(seg.map(0)[name], mine if name == "m2" else refs[0].layer(name)), name
    # the flag cleared with a slot still owing its three layers: they are still delivered, and the next cloud writes all nine
    seg.filter_cloud(batch[1], ORIGIN0, -1.73, map=seg.map(3))
    refs[3].filter_cloud(batch[1], ORIGIN0, -1.73)
    seg.set_flags(minimal_layers=False)
    assert_same_state(seg.map(3), refs[3], "after clearing the flag")
    seg.filter_cloud(batch[2], ORIGIN0, -1.73, map=seg.map(3))
    refs[3].filter_cloud(batch[2], ORIGIN0, -1.73)
    assert_same_state(seg.map(3), refs[3], "full layers again")
    seg.close()


def test_sparse_per_call_layers_at_the_host_boundary():
    """The nine per-call layers are stored sparsely (only the half columns with records of the last cloud hold values): gg_get_layer,
    the 8-bit images and the terrain image must still show the reference's dense matrices, a host write of ONE per-call layer
    (gg_set_layer) must not disturb the other eight, and clouds that move across the map must never leave stale cells behind."""
    a = synth.hdl64_cloud(seed=31, n_az=500)
    b = synth.clone_cloud(a)
    b["x"] += np.float32(17.0)   # the second cloud covers other tiles: what the first one wrote there goes stale
    b["y"] -= np.float32(9.0)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(a))
    ref = oracle.OracleMap(120.0, 0.33)
    for k, cloud in enumerate([a, b, a, synth.empty_cloud(0), b]):
        origin = (17.0, -9.0, 0.0) if cloud is b else ORIGIN0
        seg.filter_cloud(cloud, origin, -1.73)
        ref.filter_cloud(cloud, origin, -1.73)
        assert_same_state(seg.map(0), ref, f"cloud {k}")
    img, lo, hi = seg.map(0).image_u8("pointsRaw")
    raw = ref.layer("pointsRaw")
    assert lo == raw.min() and hi == raw.max() and img.shape == raw.shape and (img > 0).sum() == (raw > raw.min()).sum()
    assert np.array_equal(seg.map(0).terrain_image()[:, :, 2], raw)
    # the host overwrites one per-call layer: the other ten read back unchanged, and the next cloud starts from the per-call
    # reset values again (:61-75) whatever the host wrote
    mine = np.full(raw.shape, 7.5, dtype=np.float32)
    before = {n: seg.map(0)[n] for n in oracle.LAYERS}
    seg.map(0).set("planeDist", mine)
    for n in oracle.LAYERS:
        assert nan_equal(seg.map(0)[n], mine if n == "planeDist" else before[n]), n
    seg.filter_cloud(a, ORIGIN0, -1.73)
    ref.filter_cloud(a, ORIGIN0, -1.73)
    assert_same_state(seg.map(0), ref, "after a host write")
    seg.close()


def test_per_call_layers_round_trip_through_their_tile_blocks():
    """The per-call layers live tile by tile on the device (9 KiB blocks by Morton rank, half-column liveness masks); the host sees
    column-major matrices.  Every layer written and read back, on grids whose last tile row / column is partial (130, 364 cells)
    and on a small one (66 cells); after a cloud (sparse state), after a reset (dense state), several slots, all layers at once."""
    rng = np.random.default_rng(5)
    for length, res, slots in ((43.0, 0.33, 3), (120.0, 0.33, 2), (22.0, 0.33, 1)):
        seg = api.GroundSegmentation().init(length, res, n_slots=slots, max_points=4096)
        n = seg.map(0).getSize()[0]
        cloud = synth.random_cloud(3000, seed=9, extent=0.4 * length)
        for slot in range(slots):
            m = seg.map(slot)
            if slot % 2 == 0:
                seg.filter_cloud(cloud, ORIGIN0, -1.7, map=m)  # (sparse: only some half columns hold values)
            want = {name: rng.standard_normal((n, n)).astype(np.float32) for name in oracle.LAYERS}
            want["points"][0, 0] = np.float32("nan")
            want["variance"][n - 1, n - 1] = np.float32("inf")
            for name in oracle.LAYERS:
                m.set(name, want[name])
            for name in oracle.LAYERS:  # one at a time ...
                assert nan_equal(m[name], want[name]), (length, slot, name)
            got = m.layers()            # ... and all with one synchronisation
            for name in oracle.LAYERS:
                assert nan_equal(got[name], want[name]), (length, slot, name)
        # the other slots were not touched by the writes to this one
        if slots > 1:
            a = seg.map(0).layers()
            seg.map(1).set("m2", np.zeros((n, n), dtype=np.float32))
            b = seg.map(0).layers()
            for name in oracle.LAYERS:
                assert nan_equal(a[name], b[name]), name
        seg.close()


def test_largest_supported_grid_and_ring_group_counts():
    """The sweep's wavefronts own groups of 64 rings: 1400 x 1400 (698 rings, 11 groups: only possible as several work-groups per
    cloud), 1000 x 1000 (498 rings, 8 groups -- one work-group per group by default in a one-cloud launch, and forced onto ONE
    work-group: 3 wavefronts per side, 147 KB of LDS hand-over tables) down to grids with fewer rings than lanes; a sparse cloud
    keeps the oracle fast."""
    for length, res, gpw in ((280.0, 0.2, 0), (200.0, 0.2, 0), (200.0, 0.2, 8), (200.0, 0.2, 5), (43.0, 0.33, 0), (22.0, 0.33, 0)):
        seg = api.GroundSegmentation().init(length, res, n_slots=1, max_points=1000)
        if gpw:
            seg.debug_set_tuning("sweep_gpw", gpw)
        ref = oracle.OracleMap(length, res)
        c = synth.random_cloud(1000, seed=2, extent=0.45 * length)
        for _ in range(2):
            seg.filter_cloud(c, ORIGIN0, -1.7)
            ref.filter_cloud(c, ORIGIN0, -1.7)
            assert_same_state(seg.map(0), ref, f"{length} m / {res} m")


def test_run_to_run_determinism_and_reset():
    cloud = synth.hdl64_cloud(seed=12, order="azimuth")
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    runs = []
    for _ in range(3):
        seg.map(0).reset()
        for _ in range(2):
            _, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
        runs.append((labels.copy(), index.copy(), {k: v for k, v in seg.map(0).layers().items()}))
    for lab, idx, layers in runs[1:]:
        assert np.array_equal(lab, runs[0][0]) and np.array_equal(idx, runs[0][1])
        for k in layers:
            assert nan_equal(layers[k], runs[0][2][k]), k


def test_full_size_properties_without_oracle():
    """Size-independent properties at BASELINE's full size (also at the 2.1 M-point scale the oracle is slow on)."""
    cloud = synth.hdl64_cloud(seed=31)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    for _ in range(2):
        out, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
    cls, cell = seg.point_classes(len(cloud))
    emitted = index >= 0
    assert len(out) == emitted.sum()
    assert np.array_equal(np.sort(index[emitted]), np.arange(emitted.sum()))  # a permutation
    for k in (oracle.KEPT, oracle.IGNORED, oracle.OUTLIER):
        sel = emitted & (cls == k)
        assert (np.diff(index[sel]) > 0).all()                                 # cloud order inside each class
    assert (labels[~emitted] == 0).all() and np.isin(labels[emitted], (49, 99)).all()
    assert np.array_equal(out["intensity"], labels[emitted][np.argsort(index[emitted])].astype(np.float32))
    pts_layer = seg.map(0)["points"]
    cnt = np.bincount(cell[labels == 99], minlength=364 * 364).reshape((364, 364), order="F")
    assert np.array_equal(pts_layer, cnt.astype(np.float32))                   # :176 non-ground count per cell
    raw = seg.map(0)["pointsRaw"]
    assert raw.sum() == (cls != oracle.OUTSIDE).sum()                          # :234 every in-map point counted once


def test_capacity_and_argument_errors():
    from groundgrid_amd._lib import GroundGridError

    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=100)
    with pytest.raises(GroundGridError):
        seg.filter_cloud(synth.random_cloud(101, seed=1), ORIGIN0, -1.7)
    with pytest.raises(GroundGridError):
        api.GroundSegmentation().init(100.4, 0.4)  # GG_ERR_GEOMETRY


def test_config4_full_2M_points_one_frame():
    cloud = synth.os128_cloud(seed=2)  # ~2.1 M points, 1000 x 1000 grid (BASELINE configs[3])
    assert len(cloud) > 1_900_000
    run_pair(cloud, length=200.0, resolution=0.2, frames=1)


def test_index_fast_path_boundaries():
    """K1 replaces the f64 divide of getIndex by a multiply + exactness check (gg_device.h index_of); points that sit
    exactly on / next to cell boundaries (where the exact division must decide) have to match the oracle too."""
    m = oracle.OracleMap(120.0, 0.33)
    half, res = 0.5 * m.length[0], m.resolution
    xs = []
    for k in (0, 1, 2, 17, 181, 182, 183, 300, 362, 363):
        edge = half - k * res  # x of the boundary between rows k-1 and k (exact in double)
        f = np.float32(edge)
        xs += [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
    xs = np.array(xs, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs)
    pts = np.column_stack([X.ravel(), Y.ravel(), np.full(X.size, -1.7, dtype=np.float32)])
    run_pair(synth.make_cloud(pts), frames=1)
    # a == 0 exactly: map position chosen so that (x - L/2) - pos == 0 for x = 5 -> the exact-division path
    pos = (float(np.float64(np.float32(5.0)) - half), float(np.float64(np.float32(-7.25)) - half))
    pts = np.array([[5.0, -7.25, -1.0], [5.0, -7.0, -1.0], [4.9, -7.25, -1.2], [3.0, -9.0, -1.1]], dtype=np.float32)
    run_pair(synth.make_cloud(pts), pos=pos, origin=(pos[0], pos[1], 0.0), frames=1)


@pytest.mark.parametrize("pos", [(0.0, 0.0), (500000.3, 5800000.7), (-1.0e7, 3.3e6), (9.0e8, -9.9e8), (3.0e9, -2.0e10), (1.0e15, 1.0e15)])
def test_map_border_points_near_and_far_from_the_frame_origin(pos):
    """isInside (grid_map GridMapMath.cpp checkIfPositionWithinMap, called from src/GroundSegmentation.cpp:230) and getIndex at
    the map's four borders: points on, next to and beyond them, at UTM-sized map positions and at positions so far out that an
    ulp of the coordinates is larger than a cell, have to match the oracle.  (Written for an experiment that took the isInside
    test only for points whose index is not strictly interior -- tools/experiments/k1_interior_index_shortcut.patch: exact, no
    faster -- and kept for the cases.)"""
    m = oracle.OracleMap(120.0, 0.33, pos=pos)
    half, res = 0.5 * m.length[0], m.resolution
    offs = []
    for k in (-2, -1, 0, 1, 2, 3, 180, 361, 362, 363, 364, 365):
        for d in (0.0, 1e-7, -1e-7, 0.5 * res):
            offs.append(half - k * res + d)
    offs = np.array(offs + [1e30, -1e30, np.inf, np.nan])
    xs = (np.float64(pos[0]) + offs).astype(np.float32)
    ys = (np.float64(pos[1]) + offs).astype(np.float32)
    X, Y = np.meshgrid(xs, ys)
    pts = np.column_stack([X.ravel(), Y.ravel(), np.full(X.size, -1.7, dtype=np.float32)])
    run_pair(synth.make_cloud(pts), pos=pos, origin=(np.float32(pos[0]), np.float32(pos[1]), 0.0), frames=2)


# ---------------------------------------------------------------- N1: map follows the vehicle (GroundGrid::update)

def drive_frames(n_frames=6, n_az=500, seed=77):
    """A vehicle driving a curve: per frame (cloud in the map frame, odom xy, sensor origin, base_link<-map transform)."""
    base = synth.hdl64_cloud(seed=seed, n_az=n_az)
    frames = []
    for f in range(n_frames):
        yaw = 0.15 * f
        x, y = 1.3 * f, 0.45 * f * f * 0.3 - 0.7 * f
        c, s = np.float32(np.cos(yaw)), np.float32(np.sin(yaw))
        cloud = base.copy()
        cloud["x"] = (c * base["x"] - s * base["y"] + np.float32(x)).astype(np.float32)
        cloud["y"] = (s * base["x"] + c * base["y"] + np.float32(y)).astype(np.float32)
        cloud["z"], cloud["ring"], cloud["intensity"] = base["z"], base["ring"], base["intensity"]
        # base_link pose in map: (x, y, -1.73, yaw)  ->  base_link<-map = inverse
        tx = -(np.cos(yaw) * x + np.sin(yaw) * y)
        ty = -(-np.sin(yaw) * x + np.cos(yaw) * y)
        tf = (tx, ty, 1.73, 0.0, 0.0, -np.sin(yaw / 2), np.cos(yaw / 2))
        frames.append((cloud, (x, y), (np.float32(x), np.float32(y), np.float32(0.0)), tf))
    return frames


def test_driving_sequence_with_device_map_scroll():
    frames = drive_frames()
    n = max(len(f[0]) for f in frames)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=n)
    x0, y0 = frames[0][1]
    ref = oracle.OracleMap(120.0, 0.33, pos=(x0, y0), odom_z=-1.73)
    seg.map(0).reset(odom_z=-1.73, pos=(x0, y0))  # GroundGrid::initGroundGrid on the first odometry message
    moved_any = False
    for k, (cloud, odom, origin, tf) in enumerate(frames):
        if k > 0:  # odom_callback -> GroundGrid::update
            moved, shift = ref.update(odom[0], odom[1], tf)
            dshift = seg.map(0).move(odom[0], odom[1], tf)
            assert tuple(dshift) == tuple(shift)
            assert seg.map(0).getPosition() == ref.position
            moved_any |= moved
            for name in ("ground", "groundpatch"):
                assert nan_equal(seg.map(0)[name], ref.layer(name)), (k, name)
        out, labels, index = seg.filter_cloud(cloud, origin, -1.73, return_details=True)
        r = ref.filter_cloud(cloud, origin, -1.73)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), k
        assert out.tobytes() == r["out_points"].tobytes()
        assert_same_state(seg.map(0), ref, f"frame {k}")
    assert moved_any


def test_map_scroll_edge_cases():
    seg = api.GroundSegmentation().init(21.12, 0.33, n_slots=1, max_points=16)
    ref = oracle.OracleMap(21.12, 0.33)
    rng = np.random.default_rng(3)
    g0, w0 = rng.normal(size=(64, 64)).astype(np.float32), rng.random((64, 64)).astype(np.float32)
    for odom, tf in [((0.1, -0.1), (0, 0, 1, 0, 0, 0, 1)),            # less than half a cell: no move at all
                     ((0.17, 0.0), (0.3, 0.2, 1.5, 0.02, -0.01, 0.3, 0.95)),  # exactly past the rounding point, tilted base
                     ((-3.0, 5.2), (1, 2, 3, 0, 0, 0.7071, 0.7071)),
                     ((40.0, 5.2), (0, 0, 0.5, 0, 0, 0, 1)),               # more than the whole map in x: everything is new
                     ((40.0, -90.0), (0, 0, 0.5, 0.1, 0.2, 0.3, 0.9))]:
        ref.set_layer("ground", g0)
        ref.set_layer("groundpatch", w0)
        seg.map(0).set("ground", g0)
        seg.map(0).set("groundpatch", w0)
        moved, shift = ref.update(odom[0], odom[1], tf)
        assert tuple(seg.map(0).move(odom[0], odom[1], tf)) == tuple(shift)
        assert seg.map(0).getPosition() == ref.position
        for name in ("ground", "groundpatch"):
            assert nan_equal(seg.map(0)[name], ref.layer(name)), (odom, name)


# ---------------------------------------------------------------- N3: KITTI-format sequence replay (configs[4] harness)

def test_bench_kitti_dir_leg_on_a_kitti_format_sequence(tmp_path):
    """bench.py --kitti-dir (BASELINE configs[0] / [4] harness): replays a KITTI-format directory end to end on the device and
    prints one JSON line with clouds/s, the per-label table and its diff against the published sequence-00 table."""
    import json
    import subprocess
    import sys

    from tests.test_kitti_cpu import _synthetic_sequence

    d, _ = _synthetic_sequence(tmp_path, n_frames=4, n_az=260)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--kitti-dir", str(d), "--kitti-euler-roundtrip"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["clouds"] == 4 and r["value"] > 0 and r["table"]["clouds"] == 4
    assert "vs_readme_seq00" in r and r["vs_readme_seq00"]["within_tolerance"] is False  # 4 synthetic clouds are not sequence 00


@pytest.mark.parametrize("n_frames,n_az", [(5, 260), (70, 1200)])
def test_kitti_format_sequence_replay_matches_cpu_path(tmp_path, n_frames, n_az):
    """configs[0] / configs[4] plumbing: reader -> poses -> map scroll -> filter -> evaluator.  The long case drives 77 m
    (more than half a map width of scrolling, 0.1 rad of yaw per frame) with ~75 k points per cloud."""
    from groundgrid_amd import kitti, replay
    from tests.test_kitti_cpu import OracleBackend, _synthetic_sequence

    d, _ = _synthetic_sequence(tmp_path, n_frames=n_frames, n_az=n_az)
    seq = kitti.KittiSequence(d)
    per_frame = {"gpu": [], "cpu": []}
    ev_gpu, _ = replay.replay(seq, replay.DeviceBackend(max_points=100000), on_frame=lambda fr, lab, idx: per_frame["gpu"].append((lab.copy(), idx.copy())))
    ev_cpu, _ = replay.replay(seq, OracleBackend(), on_frame=lambda fr, lab, idx: per_frame["cpu"].append((lab.copy(), idx.copy())))
    for k, ((lg, ig), (lc, ic)) in enumerate(zip(per_frame["gpu"], per_frame["cpu"])):
        assert np.array_equal(lg, lc) and np.array_equal(ig, ic), k
    assert ev_gpu.total == ev_cpu.total and ev_gpu.non_ground == ev_cpu.non_ground
    assert ev_gpu.summary() == ev_cpu.summary() and ev_gpu.table() == ev_cpu.table()


# ---------------------------------------------------------------- N2: cloud -> map transform fused into the first kernel

def test_sensor_frame_cloud_transformed_on_device():
    import torch
    from groundgrid_amd import kitti

    base = synth.hdl64_cloud(seed=9, n_az=700)
    q = np.array([0.01, -0.02, np.sin(0.4), np.cos(0.4)])
    q /= np.linalg.norm(q)
    R, t = kitti.matrix_from_quaternion(q), np.array([3.25, -1.5, 0.07])
    cloud_map = kitti.transform_cloud(base, R, t)          # what the nodelet computes on the CPU (Nodelet.cpp:166-181)
    tf = np.hstack([R, t[:, None]])
    origin = tuple(np.float32(v) for v in t)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=2, max_points=len(base))
    ref = oracle.OracleMap(120.0, 0.33)
    for frame in range(2):
        out, labels, index = seg.filter_cloud(base, origin, -1.66, return_details=True, map_from_cloud=tf)
        r = ref.filter_cloud(cloud_map, origin, -1.66)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), frame
        assert out.tobytes() == r["out_points"].tobytes(), frame   # the returned cloud is in the map frame
        assert_same_state(seg.map(0), ref, f"frame {frame}")
    # batched entry point, PointXYZIR records, returned clouds materialised on the device
    raw = torch.from_numpy(np.frombuffer(base.tobytes(), dtype=np.uint8).reshape(1, -1, 32).copy()).cuda()
    ref2 = oracle.OracleMap(120.0, 0.33)
    o = seg.filter_batch(raw, [len(base)], [origin], [-1.66], first_slot=1, want_clouds=True, transforms=tf[None])
    torch.cuda.synchronize()
    r = ref2.filter_cloud(cloud_map, origin, -1.66)
    n_out = int(o.counts[0, 0])
    assert np.array_equal(o.labels[0, : len(base)].cpu().numpy(), r["label"])
    assert o.out_clouds[0, :n_out].cpu().numpy().tobytes() == r["out_points"].tobytes()


# ---------------------------------------------------------------- N4: wire formats around the path

def test_wire_formats_pointcloud2_ingest_and_images():
    cloud = synth.hdl64_cloud(seed=13, n_az=500)
    n = len(cloud)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=n)
    ref = oracle.OracleMap(120.0, 0.33)
    # the KITTI player's PointCloud2 payload: 18-byte records x, y, z, intensity (f32) + ring (u16)  (kitti_data_publisher.py:139-150)
    wire = np.zeros(n, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2")]))
    assert wire.dtype.itemsize == 18
    for k in ("x", "y", "z", "intensity", "ring"):
        wire[k] = cloud[k]
    for frame in range(2):
        labels, index, n_out = seg.filter_cloud_pc2(wire.tobytes(), n, 18, (0, 4, 8, 16), ORIGIN0, -1.73)
        r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]) and n_out == len(r["out_points"])
    assert_same_state(seg.map(0), ref)

    # grid_map::GridMapCvConverter::toImage<unsigned char, 1> (src/GroundGridNodelet.cpp:239), restated with numpy
    for layer in ("ground", "groundpatch", "points", "variance", "minGroundHeight"):
        img, lo, hi = seg.map(0).image_u8(layer)
        L = ref.layer(layer)
        fin = np.isfinite(L)
        elo, ehi = np.float32(L[fin].min()), np.float32(L[fin].max())
        assert (np.float32(lo), np.float32(hi)) == (elo, ehi), layer
        with np.errstate(all="ignore"):
            expect = (((np.clip(L, elo, ehi) - elo) / (ehi - elo)) * np.float32(255.0))
            expect = np.where(fin & np.isfinite(expect), expect, 0).astype(np.uint8)
        assert np.array_equal(img, expect), layer
    # the 32FC3 terrain image (src/GroundGridNodelet.cpp:247-268)
    t = seg.map(0).terrain_image()
    g, raw = ref.layer("ground"), ref.layer("pointsRaw")
    assert np.array_equal(t[:, :, 0], g) and np.array_equal(t[:, :, 2], raw)
    s = np.zeros_like(raw)
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            s[1:-1, 1:-1] += raw[1 + di : raw.shape[0] - 1 + di, 1 + dj : raw.shape[1] - 1 + dj]   # integer-valued: order-free
    assert np.array_equal(t[1:-1, 1:-1, 1], (s[1:-1, 1:-1] >= 27).astype(np.float32))


# ---------------------------------------------------------------- round 2: conventions, robustness, pipelined host call

def test_odd_grid_dense_points_on_row_n_minus_3():
    """for odd sizes the reference's quadrant split (:325-328) never runs detect_ground_patch on row n - 3.
    A 727 x 727 map (240 m / 0.33 m) with enough points on and around that row to pass the 3-point early-out."""
    n, res = 727, 0.33
    rng = np.random.default_rng(12)
    m = 60000
    # map frame: row index 0 is at +x; row r covers x in (L/2 - (r+1) res, L/2 - r res]
    L = n * np.float64(np.float32(res))
    rows = rng.integers(n - 8, n - 1, size=m)
    x = L / 2 - (rows + rng.random(m)) * np.float64(np.float32(res))
    y = rng.uniform(-L / 2 + 1, L / 2 - 1, size=m)
    z = -1.7 + rng.normal(0, 0.01, size=m)
    cloud = synth.make_cloud(np.column_stack([x, y, z]).astype(np.float32), ring=rng.integers(0, 64, m))
    r = run_pair(cloud, length=240.0, resolution=res, frames=3)
    assert (r["cls"] == oracle.KEPT).sum() > 40000


def test_eigen34_reduction_order_on_device():
    """gg_conventions.eigen_reduction = GG_EIGEN_34_SSE: the 5x5 block sums of K3 follow Eigen 3.4's slice-vectorised order;
    bit-identical to the oracle under the same switch, and different from the Eigen 3.3 terrain somewhere."""
    cloud = synth.hdl64_cloud(seed=21, n_az=900)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    grounds = {}
    try:
        for order in (1, 0):
            oracle.set_eigen_reduction(order)
            seg.set_conventions(eigen_reduction=order)
            ref = oracle.OracleMap(120.0, 0.33)
            seg.map(0).reset()
            for f in range(3):
                _, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
                r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
                assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), (order, f)
                assert_same_state(seg.map(0), ref, f"eigen order {order} frame {f}")
            grounds[order] = seg.map(0)["ground"]
    finally:
        oracle.set_eigen_reduction(0)
    assert not np.array_equal(grounds[0], grounds[1]) and np.max(np.abs(grounds[0] - grounds[1])) < 1e-3


@pytest.mark.parametrize("rotation", ["kdl", "tf2"])
def test_map_scroll_under_both_rotation_conventions(rotation):
    """gg_move_map takes matrix entries (ABI v2); whichever quaternion -> matrix convention the binding picks, device and
    oracle fill the exposed cells with the same plane."""
    seg = api.GroundSegmentation().init(21.12, 0.33, n_slots=1, max_points=16)
    ref = oracle.OracleMap(21.12, 0.33)
    q = np.array([0.013, -0.021, 0.31, 0.95])
    q /= np.linalg.norm(q)
    pose = (0.3, 0.2, 1.5) + tuple(q)
    moved, shift = ref.update(3.1, -2.2, pose, rotation=rotation)
    assert moved and tuple(seg.map(0).move(3.1, -2.2, pose, rotation=rotation)) == tuple(shift)
    for name in ("ground", "groundpatch"):
        assert nan_equal(seg.map(0)[name], ref.layer(name)), name


def test_corrupt_z_does_not_hang_the_device():
    """an in-map point with z = -1e9 walked ~1e9 line-of-sight steps.  The walk is bounded (documented
    deviation shared with the oracle); the call returns promptly and still matches."""
    import time

    good = synth.hdl64_cloud(seed=2, n_az=300)
    bad = synth.make_cloud(np.array([[4.0, 4.0, -1e9], [10.0, -3.0, -3e38], [0.2, 25.0, -1e7], [7.0, 7.0, -70000.0]], dtype=np.float32))
    cloud = synth.empty_cloud(len(good) + len(bad))
    cloud[: len(good)] = good
    cloud[len(good):] = bad
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    ref = oracle.OracleMap(120.0, 0.33)
    for f in range(3):
        t0 = time.perf_counter()
        _, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
        assert time.perf_counter() - t0 < 5.0
        r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
        cls, _ = seg.point_classes(len(cloud))
        assert np.array_equal(cls, r["cls"]) and np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), f
        assert_same_state(seg.map(0), ref, f"frame {f}")


def test_pipelined_host_call_matches_the_serial_path():
    """gg_filter_cloud_async / gg_filter_cloud_wait two clouds deep on one map: same returned clouds, labels and terrain as
    the serial oracle; a third outstanding ticket is refused."""
    clouds = [synth.hdl64_cloud(seed=30 + k, n_az=500) for k in range(3)]
    seq = [clouds[k % 3] for k in range(7)]
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=max(len(c) for c in clouds))
    ref = oracle.OracleMap(120.0, 0.33)
    tickets = [seg.filter_cloud_async(seq[0], ORIGIN0, -1.73)]
    for k in range(len(seq)):
        if k + 1 < len(seq):
            tickets.append(seg.filter_cloud_async(seq[k + 1], ORIGIN0, -1.73))
        if k == 0:
            with pytest.raises(api.GroundGridError):
                seg.filter_cloud_async(seq[0], ORIGIN0, -1.73)
        out, labels, index = seg.filter_cloud_wait(tickets[k], return_details=True)
        r = ref.filter_cloud(seq[k], ORIGIN0, -1.73)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), k
        assert out.tobytes() == r["out_points"].tobytes(), k
    assert_same_state(seg.map(0), ref, "after the pipelined sequence")


def test_batch_on_torch_default_stream_is_ordered_with_torch_ops():
    """torch's default stream handle is 0, which the C ABI reads as "the context's own stream".  The binding
    now names the legacy stream, so a torch op enqueued right behind filter_batch sees its results without any host
    synchronisation."""
    import torch

    cloud = synth.hdl64_cloud(seed=8, n_az=800)
    n = len(cloud)
    stride = (n + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=stride)
    host = np.zeros((1, stride), dtype=api.POINT16_DTYPE)
    host[0, :n] = api.pack16(cloud)
    pts = torch.from_numpy(host.view(np.uint8).reshape(1, stride, 16)).cuda()
    ref = oracle.OracleMap(120.0, 0.33)
    out = None
    for f in range(3):
        out = seg.filter_batch(pts, [n], np.zeros((1, 3), np.float32), np.array([-1.73]), out=out)
        snapshot = out.labels[0, :n].clone()          # torch op on the same (default) stream, no synchronize in between
        r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
        assert np.array_equal(snapshot.cpu().numpy(), r["label"]), f
        # ... and the library orders its own stream behind the batch: no torch.cuda.synchronize() before reading layers
        assert nan_equal(seg.map(0)["ground"], ref.layer("ground")), f


def test_reduce_tile_classes_long_cells_and_quotient_fallbacks():
    """k_reduce's corner cases: tiles with exactly 512 / 513 records (wavefront path / work-group path), a cell with more points
    than the reciprocal table holds (IEEE tail), constant heights (every delta is 0: all quotients take the exact path),
    heights so small that quotients fall below 2^-100, huge heights, and a second frame on the live tiles."""
    rng = np.random.default_rng(77)
    res = 0.33
    parts = []

    def cell_points(cx, cy, n, z):
        xy = np.column_stack([np.full(n, cx), np.full(n, cy)]) + rng.uniform(0.01, res - 0.01, size=(n, 2))
        return np.column_stack([xy, z])

    # one cell with 5000 points (> RCAP = 4080), noisy heights
    parts.append(cell_points(6 * res, 6 * res, 5000, rng.normal(-1.7, 0.03, 5000)))
    # one cell, constant height: mean == planeDist from the second point on
    parts.append(cell_points(-9 * res, 4 * res, 700, np.full(700, -1.5, np.float32)))
    # tiny and huge heights
    parts.append(cell_points(12 * res, -7 * res, 300, rng.normal(0, 1, 300) * 1e-36))
    parts.append(cell_points(-14 * res, -11 * res, 300, rng.normal(0, 1, 300) * 1e30))
    # a tile of 16x16 cells holding exactly 512 records and one holding 513 (spread over its cells)
    for tile_x, count in ((40, 512), (60, 513)):
        xy = np.column_stack([rng.uniform(tile_x * res + 0.02, (tile_x + 15) * res, count), rng.uniform(-90 * res, -76 * res, count)])
        parts.append(np.column_stack([xy, rng.normal(-1.7, 0.05, count)]))
    pts = np.concatenate(parts).astype(np.float32)
    rng.shuffle(pts)
    run_pair(synth.make_cloud(pts, ring=rng.integers(0, 64, len(pts))), frames=2)


def test_reduce_recurrence_rare_cases_off_the_fast_path():
    """k_reduce's fast recurrence assumes `mean != 0` after a cell's first point and heights that are numbers; four points
    that break either are redone with the reference's expressions.  Cells whose running mean is or returns to exactly zero,
    NaN / +-inf heights at every position of a four-point block, -0.0, in a light tile (one wavefront), in a dense tile
    (work-group, count-sorted lanes) and in a tile whose fullest cells run one chain per wavefront."""
    rng = np.random.default_rng(123)
    res = 0.33
    seqs = [
        [0.0, 0.0, 0.5, -0.5, 0.25, 0.0, 1.0],
        [1.0, -1.0, 3.0, 0.125, -0.125],          # the mean returns to exactly 0 after the second point
        [-0.0, 2.0, -2.0, -0.0, 0.0, 7.0],
        [2.0, 2.0, -4.0, 1.0, 1.0, 1.0, 1.0, -4.0],
        [-1.7, np.inf, -1.6, -np.inf, -1.5, -1.4],
    ]
    for k in range(9):                               # a NaN at position k
        z = list(rng.normal(-1.7, 0.02, 12))
        z[k] = np.nan
        seqs.append(z)
    long_cell = list(rng.normal(-1.7, 0.02, 40))    # >= 24 points: the tile's fullest cells run one chain per wavefront
    long_cell[17] = 0.0
    long_cell[23] = np.nan
    long_cell[24] = 0.0
    long_cell[25] = 0.0
    seqs.append(long_cell)

    def region(x0, y0, fillers):
        """the sequences in cells (x0 + k, y0) ... of one 16x16 tile, `fillers` more points spread over the tile's other rows"""
        parts = []
        for k, z in enumerate(seqs):
            cx, cy = (x0 + k % 16) * res, (y0 + k // 16) * res
            xy = np.column_stack([np.full(len(z), cx), np.full(len(z), cy)]) + rng.uniform(0.02, res - 0.02, size=(len(z), 2))
            parts.append(np.column_stack([xy, np.asarray(z, np.float64)]))
        if fillers:
            xy = np.column_stack([rng.uniform(x0 * res + 0.02, (x0 + 15) * res, fillers), rng.uniform((y0 + 3) * res, (y0 + 12) * res, fillers)])
            parts.append(np.column_stack([xy, rng.normal(-1.7, 0.05, fillers)]))
        return np.concatenate(parts)

    # (a region straddles up to four tiles: enough fillers that its tiles leave the single-wavefront path / split their chains)
    pts = np.concatenate([region(30, 30, 0), region(-70, 30, 3000), region(30, -70, 12000)]).astype(np.float32)
    run_pair(synth.make_cloud(pts, ring=rng.integers(0, 64, len(pts))), frames=2)


@pytest.mark.parametrize("mdf,thres,obs", [(0.0005, 0.3, 0.1), (2e-5, 0.3, 0.1), (2e-6, 0.3, 0.1), (1e-7, 0.3, 0.1), (0.0005, 0.1, 0.3),
                                           (0.0005, 0.0, 0.1), (-0.0005, 0.3, 0.1), (0.02, 0.3, 0.299)])
def test_label_tolerance_branches(mdf, thres, obs):
    """k_label decides the clamp of the tolerance (:170-171) from the point's CELL (its distance from the origin to within
    0.75 cell) and reads x, y only when the two ends of that bound disagree.  Factors that put most points above the clamp,
    inside the band, below it; thresholds in the unusual order, zero, a negative factor; origin off the map centre."""
    def edit(c):
        c.minimum_distance_factor = mdf
        c.miminum_point_height_threshold = thres
        c.minimum_point_height_obstacle_threshold = obs
    r = run_pair(synth.hdl64_cloud(seed=33, n_az=500), origin=(3.7, -2.2, 0.1), frames=2, cfg_edit=edit)
    assert len(np.unique(r["label"])) >= 2


def test_label_tolerance_band_in_the_sensor_frame():
    """the same with the cloud handed over in the sensor frame (x, y of the exact branch come from the transform)"""
    from groundgrid_amd import kitti

    base = synth.hdl64_cloud(seed=34, n_az=500)
    q = np.array([0.02, 0.01, np.sin(-0.3), np.cos(-0.3)])
    q /= np.linalg.norm(q)
    R, t = kitti.matrix_from_quaternion(q), np.array([-2.5, 4.0, 0.05])
    cloud_map = kitti.transform_cloud(base, R, t)
    tf = np.hstack([R, t[:, None]])
    origin = tuple(np.float32(v) for v in t)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(base))
    ref = oracle.OracleMap(120.0, 0.33)
    ref.cfg.minimum_distance_factor = 2e-6
    c = seg.getConfig()
    c.minimum_distance_factor = 2e-6
    seg.setConfig(c)
    for frame in range(2):
        out, labels, index = seg.filter_cloud(base, origin, -1.66, return_details=True, map_from_cloud=tf)
        r = ref.filter_cloud(cloud_map, origin, -1.66)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), frame
        assert out.tobytes() == r["out_points"].tobytes(), frame
        assert_same_state(seg.map(0), ref, f"frame {frame}")
    seg.close()


def test_reset_maps_on_the_callers_stream_is_ordered_with_batches():
    """gg_reset_maps(..., stream) between batches on the same torch stream: no event hand-over to the context's stream, same
    results as a fresh context (cold maps every step, as bench.py's headline does)."""
    import torch

    clouds = [synth.hdl64_cloud(seed=31 + k, n_az=600) for k in range(3)]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=3, max_points=stride)
    host = np.zeros((3, stride), dtype=api.POINT16_DTYPE)
    n = []
    for b, c in enumerate(clouds):
        host[b, :len(c)] = api.pack16(c)
        n.append(len(c))
    pts = torch.from_numpy(host.view(np.uint8).reshape(3, stride, 16)).cuda()
    org = np.zeros((3, 3), np.float32)
    bz = np.full(3, -1.73)
    side = torch.cuda.Stream()
    out = None
    with torch.cuda.stream(side):
        for step in range(3):  # warm the maps, then re-initialise them on the same stream and run again: the last step is cold
            if step == 2:
                seg.reset_maps(0, 3, odom_z=0.0, persistent_only=True, on_torch_stream=True)
            out = seg.filter_batch(pts, n, org, bz, out=out)
    side.synchronize()
    labels = out.labels.cpu().numpy()
    for b, c in enumerate(clouds):
        ref = oracle.OracleMap(120.0, 0.33)
        r = ref.filter_cloud(c, (0.0, 0.0, 0.0), -1.73)
        assert np.array_equal(labels[b, :len(c)], r["label"]), f"cloud {b}"
        assert nan_equal(seg.map(b)["ground"], ref.layer("ground")), f"cloud {b} ground"
    seg.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_fuzz(seed):
    """Random geometry, random clusters (one of them packing tens of thousands of points into a single 16x16 tile), random map
    position / origin / base height, three frames: everything compared with the oracle."""
    rng = np.random.default_rng(1000 + seed)
    # (GroundSegmentation::init takes the dimension as size_t: whole metres)
    length, resolution = [(20.0, 0.2), (30.0, 0.25), (40.0, 0.33), (50.0, 0.5), (64.0, 0.33), (80.0, 0.33), (45.0, 0.2), (33.0, 0.25)][
        int(rng.integers(0, 8))]
    n_clusters = int(rng.integers(3, 9))
    parts = []
    for k in range(n_clusters):
        centre = rng.uniform(-0.55 * length, 0.55 * length, size=2)  # some clusters straddle or miss the map
        spread = float(rng.choice([0.05, 0.3, 1.5, 6.0, 20.0]))
        m = int(rng.integers(50, 40000 if k == 0 else 6000))
        xy = centre + rng.normal(0, spread, size=(m, 2))
        z = rng.normal(rng.uniform(-2.5, 0.5), rng.choice([0.0, 0.02, 0.4]), size=m)
        parts.append(np.column_stack([xy, z]))
    pts = np.concatenate(parts).astype(np.float32)
    rng.shuffle(pts)
    pos = tuple(np.round(rng.uniform(-3, 3, size=2), 2))
    origin = (float(pos[0]) + float(rng.uniform(-1, 1)), float(pos[1]) + float(rng.uniform(-1, 1)), float(rng.uniform(-0.2, 0.2)))
    cloud = synth.make_cloud(pts + np.array([pos[0], pos[1], 0.0], np.float32), ring=rng.integers(0, 64, len(pts)))
    run_pair(cloud, length=length, resolution=resolution, pos=pos, origin=origin, base_z=float(rng.uniform(-2.0, -1.4)), frames=3,
             odom_z=float(rng.uniform(-0.5, 0.5)))


@pytest.mark.parametrize("wgs,waves", [(2, 0), (1, 0), (2, 1), (2, 3), (1, 2)])
@pytest.mark.parametrize("length,resolution,batch", [(4.0, 0.33, 2), (22.0, 0.33, 3), (23.0, 0.33, 1), (61.0, 0.25, 5), (120.0, 0.33, 1), (120.0, 0.33, 16), (240.0, 0.33, 2)])
def test_pair_sweep_shapes(length, resolution, batch, wgs, waves):
    """Launches of at most 16 clouds sweep with k_sweep_pair (sweep_pair.h: both sides of a ring hand-over in one wavefront, OLD values
    from k_sweep_records, heights through a result stream and k_sweep_finish): one work-group per cloud with both pairs of sides or two
    with one pair each, one wavefront per 32-ring group or fewer (a wavefront then takes several groups in turn), maps from a single
    ring group (12 x 12) to 23 groups (727 x 727: the 128-register variant of the kernel).  Three frames, labels of every cloud and all
    layers of some against the oracle."""
    import torch

    clouds = _rotated_clouds(batch, length)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
    seg.debug_set_tuning("sweep_pair_wgs", wgs)
    seg.debug_set_tuning("sweep_pair_waves", waves)
    refs = [oracle.OracleMap(length, resolution) for _ in clouds]
    pts = _batch_inputs(16, clouds, stride)
    out = None
    for frame in range(3):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((batch, 3), np.float32), np.full(batch, -1.73 + 0.01 * frame), out=out)
        torch.cuda.synchronize()
        labels = out.labels.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, ORIGIN0, -1.73 + 0.01 * frame)
            assert np.array_equal(labels[b, : len(c)], r["label"]), (frame, b)
            if b in (0, 1, batch // 2, batch - 1):
                assert_same_state(seg.map(b), refs[b], f"frame {frame} cloud {b}")
    seg.close()


@pytest.mark.parametrize("waves", [0, 1, 2])
@pytest.mark.parametrize("length,resolution,batch", [(4.0, 0.33, 2), (22.0, 0.33, 3), (23.0, 0.33, 1), (50.0, 0.33, 24), (61.0, 0.25, 5), (120.0, 0.33, 2), (240.0, 0.33, 1)])
def test_throughput_pair_sweep_shapes(length, resolution, batch, waves):
    """Launches of more than 16 clouds sweep with k_sweep_pair_batch (sweep_pairb.h: the pair sweep on the layer in place, one work-group
    per cloud); here it takes every launch (tuning sweep_pair = 4), with up to three wavefronts per pair of sides or fewer, on maps from a
    single ring group to 23.  Three frames, labels of every cloud and all layers of some against the oracle."""
    import torch

    clouds = _rotated_clouds(batch, length)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
    seg.debug_set_tuning("sweep_pair", 4)
    seg.debug_set_tuning("sweep_pair_waves", waves)
    refs = [oracle.OracleMap(length, resolution) for _ in clouds]
    pts = _batch_inputs(16, clouds, stride)
    out = None
    for frame in range(3):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((batch, 3), np.float32), np.full(batch, -1.73 + 0.01 * frame), out=out)
        torch.cuda.synchronize()
        labels = out.labels.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, ORIGIN0, -1.73 + 0.01 * frame)
            assert np.array_equal(labels[b, : len(c)], r["label"]), (frame, b)
            if b in (0, 1, batch // 2, batch - 1):
                assert_same_state(seg.map(b), refs[b], f"frame {frame} cloud {b}")
    seg.close()


def _reset_persistent(ref, odom_z):
    """gg_reset_maps(persistent_only) on the oracle's map: the two layers that outlive a cloud take GroundGrid.cpp:71-75's values."""
    ref.set_layer("ground", np.full((ref.rows, ref.cols), np.float32(odom_z)))
    ref.set_layer("groundpatch", np.full((ref.rows, ref.cols), np.float32(0.0000001)))


def _fresh_batch(length, resolution, batch, n_az, halves=False):
    """batch maps, three calls: on fresh maps (the reset left their interior unwritten), on the maps that call left (warm), and after a
    second reset (persistent state only, another height) -- labels of every cloud, all layers of some, against the oracle."""
    import torch

    clouds = _rotated_clouds(batch, length, n_az=n_az)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
    if halves:
        seg.set_flags(concurrent_halves=True)
    refs = [oracle.OracleMap(length, resolution, odom_z=0.25) for _ in clouds]
    seg.reset_maps(odom_z=0.25)
    pts = _batch_inputs(16, clouds, stride)
    out = None
    watched = sorted({0, 1, batch // 2 - 1, batch // 2, batch - 1})
    for call in range(3):
        if call == 2:
            seg.reset_maps(odom_z=-0.5, persistent_only=True, on_torch_stream=True)
            for r in refs:
                _reset_persistent(r, -0.5)
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((batch, 3), np.float32), np.full(batch, -1.73 + 0.01 * call), out=out)
        if halves:
            seg.batch_fence()
        torch.cuda.synchronize()
        labels = out.labels.cpu().numpy()
        for b, c in enumerate(clouds):
            r = refs[b].filter_cloud(c, ORIGIN0, -1.73 + 0.01 * call)
            assert np.array_equal(labels[b, : len(c)], r["label"]), (call, b)
            if b in watched:
                assert_same_state(seg.map(b), refs[b], f"call {call} cloud {b}")
            elif call != 1:  # the two layers the fresh path is about, of EVERY map it swept (a cell k_patch wrote in the ring no sweep visits
                for name in ("ground", "groundpatch"):  # -- maps with an odd number of rows -- once got the reset's pair back: 36 of 260 maps)
                    assert nan_equal(seg.map(b)[name], refs[b].layer(name)), (call, b, name)
    seg.close()


@pytest.mark.parametrize("length,resolution,batch,n_az", [(22.0, 0.33, 260, 60), (61.0, 0.25, 258, 100), (61.0, 0.25, 130, 100), (120.0, 0.33, 257, 120), (120.0, 0.33, 100, 120), (200.0, 0.2, 40, 200)])
def test_fresh_maps_are_swept_as_they_are(length, resolution, batch, n_az):
    """gg_reset_maps leaves the interior of the (ground, confidence) layer unwritten (only the never-swept border, one padding element with
    the reset's pair and the written-cell bits); a batch of such maps that k_sweep takes without split steps -- one work-group per cloud,
    or several (364 x 364: 100 clouds in two parts each; 1000 x 1000: 40 clouds in three or four) -- runs k_patch and k_sweep in their
    FRESH variants: one ring group (66 x 66), two (244 x 244), three with a partial last one (364 x 364), eight (1000 x 1000)."""
    _fresh_batch(length, resolution, batch, n_az)


def test_fresh_maps_in_concurrent_halves():
    _fresh_batch(22.0, 0.33, 520, 60, halves=True)


def test_fresh_maps_read_or_mixed_with_warm_ones_are_filled_first():
    """Anything but a large all-fresh launch fills a fresh map before touching it: the getters return the reset's values, a launch that
    mixes fresh and warm maps sees both right, and so does a single cloud."""
    import torch

    length, resolution, batch = 22.0, 0.33, 260
    clouds = _rotated_clouds(batch, length, n_az=60)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
    refs = [oracle.OracleMap(length, resolution, odom_z=0.125) for _ in clouds]
    seg.reset_maps(odom_z=0.125)
    g = seg.map(3)["ground"]
    assert np.all(g == np.float32(0.125)) and np.all(seg.map(3)["groundpatch"] == np.float32(0.0000001))
    assert_same_state(seg.map(batch - 1), refs[batch - 1], "fresh, never filtered")
    pts = _batch_inputs(16, clouds, stride)
    n = [len(c) for c in clouds]
    zero = np.zeros((batch, 3), np.float32)
    out = seg.filter_batch(pts, n, zero, np.full(batch, -1.73))
    torch.cuda.synchronize()
    for b, c in enumerate(clouds):
        refs[b].filter_cloud(c, ORIGIN0, -1.73)
    # every second pair of maps re-initialised: the next launch holds fresh and warm maps
    for b in range(0, batch, 4):
        seg.reset_maps(first_slot=b, n_slots=2, odom_z=-0.25, persistent_only=True, on_torch_stream=True)
        _reset_persistent(refs[b], -0.25)
        _reset_persistent(refs[b + 1], -0.25)
    out = seg.filter_batch(pts, n, zero, np.full(batch, -1.7), out=out)
    torch.cuda.synchronize()
    labels = out.labels.cpu().numpy()
    for b, c in enumerate(clouds):
        r = refs[b].filter_cloud(c, ORIGIN0, -1.7)
        assert np.array_equal(labels[b, : len(c)], r["label"]), b
    for b in (0, 1, 2, 3, batch - 1):
        assert_same_state(seg.map(b), refs[b], f"mixed launch, cloud {b}")
    # a single cloud on a fresh map
    seg.reset_maps(first_slot=5, n_slots=1, odom_z=0.5)
    ref = oracle.OracleMap(length, resolution, odom_z=0.5)
    _, labels1, _ = seg.filter_cloud(clouds[5], ORIGIN0, -1.73, map=seg.map(5), return_details=True)
    assert np.array_equal(labels1, ref.filter_cloud(clouds[5], ORIGIN0, -1.73)["label"])
    assert_same_state(seg.map(5), ref, "single cloud on a fresh map")
    seg.close()


def test_fresh_maps_switched_off_write_every_cell():
    """tuning fresh_maps = 0: gg_reset_maps fills the layer as before; the results are the same."""
    import torch

    length, resolution, batch = 22.0, 0.33, 260
    clouds = _rotated_clouds(batch, length, n_az=60)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=batch, max_points=stride)
    seg.debug_set_tuning("fresh_maps", 0)
    seg.reset_maps(odom_z=0.25)
    pts = _batch_inputs(16, clouds, stride)
    out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((batch, 3), np.float32), np.full(batch, -1.73))
    torch.cuda.synchronize()
    labels = out.labels.cpu().numpy()
    for b in (0, 7, batch - 1):
        ref = oracle.OracleMap(length, resolution, odom_z=0.25)
        r = ref.filter_cloud(clouds[b], ORIGIN0, -1.73)
        assert np.array_equal(labels[b, : len(clouds[b])], r["label"]), b
        assert_same_state(seg.map(b), ref, f"cloud {b}")
    seg.close()


def test_pair_sweep_and_k_sweep_leave_the_same_map():
    """The two sweeps are interchangeable launch by launch: a map swept alternately by either (the pair sweep switched off for every
    other frame) stays bit-identical to the oracle's."""
    cloud = synth.hdl64_cloud(seed=77, n_az=900)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    ref = oracle.OracleMap(120.0, 0.33)
    for frame in range(6):
        seg.debug_set_tuning("sweep_pair", 2 if frame % 2 else 0)
        _, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
        r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), frame
        assert_same_state(seg.map(0), ref, f"frame {frame}")
    seg.close()


def test_synthetic_drive_of_1200_frames_matches_the_cpu_path_to_the_last_map():
    """BASELINE configs[4]'s shape (thousands of consecutive clouds on one scrolling map; the dataset is absent): 1200 frames of
    groundgrid_amd.kitti.synthetic_drive -- a closed loop of 0.8 m per frame, the map scrolling two or three cells in every frame,
    ~45 k points per cloud -- through the device path and the oracle side by side.  Labels and returned-cloud order in every frame,
    ground / groundpatch after the last (confidence decay and scroll seeding accumulate over a drive: 70 frames do not show a drift),
    the evaluator's table."""
    from groundgrid_amd import kitti, replay
    from tests.test_kitti_cpu import OracleBackend

    n = 1200
    dev_b, cpu_b = replay.DeviceBackend(max_points=60000), OracleBackend()
    ev, t_dev, t_cpu, n_cpu, same, first_bad = replay.replay_side_by_side(kitti.synthetic_drive(n, n_scenes=5, n_az=760), dev_b, cpu_b)
    assert same, f"labels / order differ first in frame {first_bad}"
    assert n_cpu == n and ev.cloud_count == n
    for name in ("ground", "groundpatch", "points", "variance"):
        assert nan_equal(dev_b.map.get(name), cpu_b.m.layer(name)), name
    # the loop closes: the map is back within a metre of where it started, after scrolling ~960 m
    assert np.hypot(*dev_b.map.getPosition()) < 1.5
    dev_b.seg.close()
