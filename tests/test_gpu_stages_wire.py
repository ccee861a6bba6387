"""GPU parity (pytest -m gpu) of what round 5 added around the path:

* the public stage members of the reference's class one by one (include/groundgrid/GroundSegmentation.h:59-62) through
  gg_run_stage, on layers that were NOT produced by the insert kernels, against the oracle's ggo_stage_* / single-cell twins;
* the emission side of the wire formats: the returned cloud as 18-byte PointCloud2 records written by the label
  kernel, and the serialised grid_map_msgs/GridMap.

Bit-exact throughout (NaN == NaN).  Nothing here reads /root/reference.
"""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from groundgrid_amd import api, synth  # noqa: E402
from oracle import oracle  # noqa: E402

ORIGIN0 = (0.0, 0.0, 0.0)


def nan_equal(a, b):
    return np.array_equal(a, b, equal_nan=True)


def assert_layers(seg_map, ref, names, tag=""):
    for name in names:
        a, b = seg_map[name], ref.layer(name)
        if not nan_equal(a, b):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            raise AssertionError(f"{tag} layer {name}: {len(bad)} cells differ, first {bad[:3].tolist()} "
                                 f"gpu={[float(a[tuple(i)]) for i in bad[:3]]} ref={[float(b[tuple(i)]) for i in bad[:3]]}")


def synthetic_layers(n, seed, integer_points=True):
    """Layer contents no cloud produced: patchy counts, heights on a slope with steps, small and large variances, old terrain with
    confidences on both sides of every threshold of :379-393."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    pts = rng.integers(0, 30, (n, n)).astype(np.float32) * (rng.random((n, n)) < 0.45)
    if not integer_points:
        pts = (pts * rng.uniform(0.5, 1.5, (n, n))).astype(np.float32)
    slope = (-1.7 + 0.01 * (ii - n / 2) + 0.2 * np.sin(jj / 9.0)).astype(np.float32)
    mn = np.where(pts > 0, slope + rng.normal(0, 0.02, (n, n)).astype(np.float32), np.float32(3.402823466e+38)).astype(np.float32)
    m2 = (pts * rng.choice(np.array([1e-5, 3e-4, 2e-2], dtype=np.float32), (n, n)) * rng.random((n, n)).astype(np.float32)).astype(np.float32)
    ground = (slope + rng.normal(0, 0.3, (n, n))).astype(np.float32)
    patch = rng.choice(np.array([0.0, 1e-7, 0.2, 0.45, 0.55, 0.9, 1.0], dtype=np.float32), (n, n))
    return dict(points=pts, minGroundHeight=mn, m2=m2, ground=ground, groundpatch=patch.astype(np.float32))


def pair_with_layers(length, resolution, layers):
    seg = api.GroundSegmentation().init(length, resolution, n_slots=1, max_points=64)
    ref = oracle.OracleMap(length, resolution)
    for name, arr in layers.items():
        seg.map(0).set(name, arr)
        ref.set_layer(name, arr)
    return seg, ref


@pytest.mark.parametrize("length,resolution,integer_points", [(120.0, 0.33, True), (33.0, 0.33, False), (61.0, 0.25, True)])
def test_stage_detect_ground_patches_by_section_on_foreign_layers(length, resolution, integer_points):
    n = oracle.OracleMap(length, resolution).rows
    seg, ref = pair_with_layers(length, resolution, synthetic_layers(n, seed=n, integer_points=integer_points))
    for section in (2, 0, 3, 1):  # any order: a quadrant writes only its own cells
        seg.map(0).detect_ground_patches(section)
        ref.stage_detect_section(section)
        assert_layers(seg.map(0), ref, ("ground", "groundpatch", "variance", "points", "minGroundHeight", "m2"), f"section {section}")
    # ... and all four at once on what that left (second pass: old confidences now above 0.5 in many cells)
    seg.map(0).detect_ground_patches(-1)
    ref.stage_detect()
    assert_layers(seg.map(0), ref, ("ground", "groundpatch", "variance"), "all sections")
    seg.close()


def test_stage_detect_with_the_other_eigen_order():
    n = oracle.OracleMap(61.0, 0.25).rows
    seg, ref = pair_with_layers(61.0, 0.25, synthetic_layers(n, seed=5, integer_points=False))
    seg.set_conventions(eigen_reduction=1)
    oracle.set_eigen_reduction(1)
    try:
        seg.map(0).detect_ground_patches(-1)
        ref.stage_detect()
        assert_layers(seg.map(0), ref, ("ground", "groundpatch", "variance"))
    finally:
        oracle.set_eigen_reduction(0)
    seg.close()


@pytest.mark.parametrize("length,resolution", [(120.0, 0.33), (10.0, 0.5), (150.0, 0.25)])
def test_stage_spiral_ground_interpolation_on_foreign_layers(length, resolution):
    n = oracle.OracleMap(length, resolution).rows
    layers = synthetic_layers(n, seed=3 * n)
    seg, ref = pair_with_layers(length, resolution, layers)
    for base_z in (-1.73, 0.25):
        seg.map(0).spiral_ground_interpolation(base_z)
        ref.stage_spiral(base_z)
        # the stage is not filter_cloud: `points` keeps its counts (:147 is the caller's)
        assert_layers(seg.map(0), ref, ("ground", "groundpatch", "points"), f"base_z {base_z}")
    seg.close()


def test_stage_after_a_cloud_then_filter_again():
    """The stages work on what a cloud left (sparse per-call layers, K2's tile lists) and the next cloud works on what they left."""
    cloud = synth.hdl64_cloud(seed=21, n_az=400)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    ref = oracle.OracleMap(120.0, 0.33)
    seg.filter_cloud(cloud, ORIGIN0, -1.73)
    ref.filter_cloud(cloud, ORIGIN0, -1.73)
    seg.map(0).detect_ground_patches(1)
    ref.stage_detect_section(1)
    seg.map(0).spiral_ground_interpolation(-1.5)
    ref.stage_spiral(-1.5)
    for name in oracle.LAYERS:
        assert nan_equal(seg.map(0)[name], ref.layer(name)), name
    _, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, return_details=True)
    r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
    assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"])
    for name in oracle.LAYERS:
        assert nan_equal(seg.map(0)[name], ref.layer(name)), name
    seg.close()


def test_single_cell_stages():
    n = oracle.OracleMap(33.0, 0.33).rows
    layers = synthetic_layers(n, seed=77, integer_points=False)
    seg, ref = pair_with_layers(33.0, 0.33, layers)
    seg.map(0).detect_ground_patches(0)  # (gives `variance` contents)
    ref.stage_detect_section(0)
    rng = np.random.default_rng(1)
    for _ in range(40):
        S = int(rng.choice([3, 5]))
        i, j = (int(v) for v in rng.integers(S // 2, n - S // 2, 2))
        seg.map(0).detect_ground_patch(S, i, j)
        ref.detect_ground_patch(S, i, j)
        x, y = (int(v) for v in rng.integers(1, n - 1, 2))
        seg.map(0).interpolate_cell(x, y)
        ref.interpolate_cell(x, y)
    assert_layers(seg.map(0), ref, ("ground", "groundpatch"))
    # blocks that would leave the map are UB in the reference and an error here
    with pytest.raises(api.GroundGridError):
        seg.map(0).detect_ground_patch(5, 1, 10)
    with pytest.raises(api.GroundGridError):
        seg.map(0).interpolate_cell(0, 3)
    with pytest.raises(api.GroundGridError):
        seg.map(0).detect_ground_patches(4)
    seg.close()


# ---------------------------------------------------------------- N4: the emission side

def expected_pc2(out_points):
    return api.to_pc2(out_points).tobytes()


@pytest.mark.parametrize("with_tf", [False, True])
def test_returned_cloud_as_18_byte_pointcloud2_records(with_tf):
    cloud = synth.hdl64_cloud(seed=13, n_az=500)
    low = synth.clone_cloud(cloud)
    rng = np.random.default_rng(2)
    sel = rng.random(len(low)) < 0.1
    low["z"][sel] -= np.float32(1.0)  # outlier candidates for the second frame: all three parts of the returned cloud
    n = len(cloud)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=n)
    ref = oracle.OracleMap(120.0, 0.33)
    tf = None
    if with_tf:  # the payload is still in the sensor frame: rotate / shift it back on the host for the oracle's input
        tf = api.transform_from_pose((1.5, -0.5, 0.2, 0.0, 0.0, np.sin(0.15), np.cos(0.15)))
    n_outliers = 0
    for frame, c in enumerate((cloud, low, low)):
        wire = api.to_pc2(c)
        ref_in = c
        if with_tf:
            ref_in = synth.clone_cloud(c)
            R, t = tf[:, :3], tf[:, 3]
            p = np.stack([c["x"], c["y"], c["z"]], 1).astype(np.float64)
            for k, name in enumerate(("x", "y", "z")):  # tf2::doTransform: dot products left to right in double, cast to float
                ref_in[name] = (((R[k, 0] * p[:, 0] + R[k, 1] * p[:, 1]) + R[k, 2] * p[:, 2]) + t[k]).astype(np.float32)
        out = seg.filter_cloud_pc2_out(wire.tobytes(), n, 18, (0, 4, 8, 16), ORIGIN0, -1.73, map_from_cloud=tf)
        r = ref.filter_cloud(ref_in, ORIGIN0, -1.73)
        assert len(out) == len(r["out_points"]), frame
        assert out.tobytes() == expected_pc2(r["out_points"]), frame
        n_outliers += int((r["cls"] == oracle.OUTLIER).sum())
    assert n_outliers > 0
    for name in ("ground", "groundpatch", "points"):
        assert nan_equal(seg.map(0)[name], ref.layer(name)), name
    seg.close()


def test_pc2_records_from_the_batched_call_both_point_formats():
    import torch

    clouds = [synth.hdl64_cloud(seed=40 + k, n_az=200 + 37 * k) for k in range(3)] + [synth.empty_cloud(0)]
    stride = max(len(c) for c in clouds) + 7
    B = len(clouds)
    for rec in (16, 32):
        seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
        pts = np.zeros((B, stride, rec), dtype=np.uint8)
        for b, c in enumerate(clouds):
            src = api.pack16(c) if rec == 16 else c
            pts[b, : len(c)] = np.frombuffer(src.tobytes(), dtype=np.uint8).reshape(len(c), rec)
        d = torch.from_numpy(pts).cuda()
        o = seg.filter_batch(d, [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73), want_pc2=True)
        torch.cuda.synchronize()
        counts = o.counts.cpu().numpy()
        raw = o.out_pc2.cpu().numpy()
        for b, c in enumerate(clouds):
            ref = oracle.OracleMap(120.0, 0.33)
            r = ref.filter_cloud(c, ORIGIN0, -1.73)
            assert counts[b, 0] == len(r["out_points"])
            assert raw[b, : counts[b, 0] * 18].tobytes() == expected_pc2(r["out_points"]), (rec, b)
        seg.close()


class Reader:
    def __init__(self, buf):
        self.b, self.at = buf, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.at)
        self.at += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string(self):
        n = self.take("I")
        s = self.b[self.at : self.at + n].decode()
        self.at += n
        return s


def parse_gridmap(buf):
    """grid_map_msgs/GridMap, ROS 1 serialisation (the message definition of grid_map_msgs 1.6.x)."""
    r = Reader(buf)
    m = dict(seq=r.take("I"), stamp=r.take("II"), frame_id=r.string(), resolution=r.take("d"), length=r.take("dd"), position=r.take("ddd"),
             orientation=r.take("dddd"))
    m["layers"] = [r.string() for _ in range(r.take("I"))]
    m["basic_layers"] = [r.string() for _ in range(r.take("I"))]
    m["data"] = []
    for _ in range(r.take("I")):
        dims = [(r.string(), r.take("I"), r.take("I")) for _ in range(r.take("I"))]
        offset = r.take("I")
        count = r.take("I")
        arr = np.frombuffer(buf, dtype="<f4", count=count, offset=r.at)
        r.at += 4 * count
        m["data"].append((dims, offset, arr))
    m["start"] = r.take("HH")
    assert r.at == len(buf)
    return m


def test_gridmap_message_payload():
    cloud = synth.hdl64_cloud(seed=31, n_az=300)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=2, max_points=len(cloud))
    ref = oracle.OracleMap(120.0, 0.33, pos=(3.3, -6.6))
    seg.map(1).reset(pos=(3.3, -6.6))
    seg.filter_cloud(cloud, (3.0, -6.0, 0.0), -1.73, map=seg.map(1))
    ref.filter_cloud(cloud, (3.0, -6.0, 0.0), -1.73)
    raw = seg.map(1).gridmap_message(seq=7, stamp=(1234, 5678))
    # the layout the pinning kit holds against a real grid_map_ros (tools/pin/compare.py gridmap_message_bytes) is the library's
    import importlib.util
    spec = importlib.util.spec_from_file_location("pin_compare", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pin", "compare.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    assert raw == pin.gridmap_message_bytes(ref.rows, ref.cols, ref.resolution, ref.length, (3.3, -6.6), [(k, ref.layer(k)) for k in oracle.LAYERS], (1234, 5678), seq=7)
    m = parse_gridmap(raw)
    assert (m["seq"], m["stamp"], m["frame_id"]) == (7, (1234, 5678), "map")
    assert m["resolution"] == ref.resolution and m["length"] == ref.length and m["position"] == (3.3, -6.6, 0.0) and m["orientation"] == (0.0, 0.0, 0.0, 1.0)
    assert m["layers"] == oracle.LAYERS and m["basic_layers"] == [] and m["start"] == (0, 0)
    n = ref.rows
    for name, (dims, offset, arr) in zip(m["layers"], m["data"]):
        assert dims == [("column_index", n, n * n), ("row_index", n, n)] and offset == 0
        assert nan_equal(arr.reshape((n, n), order="F"), ref.layer(name)), name
    # a subset, with basic layers, as a publisher with two subscribers would ask for it
    m = parse_gridmap(seg.map(1).gridmap_message(layers=["ground", "variance"], frame_id="odom", basic_layers=["ground"]))
    assert m["layers"] == ["ground", "variance"] and m["basic_layers"] == ["ground"] and m["frame_id"] == "odom"
    assert nan_equal(m["data"][1][2].reshape((n, n), order="F"), ref.layer("variance"))
    seg.close()


# ---------------------------------------------------------------- the reference call shape: graph replay, fused filter + layers

def test_one_cloud_per_call_replays_a_captured_graph_100_times():
    """the single-slot sequence as a HIP graph; replayed 100 times against the oracle, with clouds of different
    sizes (the captured grids cover the buffer's capacity), a configuration change in between (captures are dropped) and eager
    calls mixed in."""
    clouds = [synth.hdl64_cloud(seed=60 + k, n_az=150 + 53 * k) for k in range(5)] + [synth.empty_cloud(0), synth.hdl64_cloud(seed=70, n_az=40)[:37]]
    cap = max(len(c) for c in clouds) + 1000
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=2, max_points=cap)
    ref = oracle.OracleMap(120.0, 0.33)
    seg.debug_set_tuning("graphs", 1)  # (opt-in: GG_GRAPH=1)
    before = seg.debug_set_tuning("graph_replays", 0)
    for it in range(100):
        c = clouds[it % len(clouds)]
        if it == 40:  # a new configuration: what was captured carries the old one by value
            cfg = seg.getConfig()
            cfg.max_ring = 50
            cfg.outlier_tolerance = 0.15
            seg.setConfig(cfg)
            ref.cfg.max_ring = 50
            ref.cfg.outlier_tolerance = 0.15
        if it == 70:
            seg.debug_set_tuning("graphs", 0)
        if it == 75:
            seg.debug_set_tuning("graphs", 1)
        out, labels, index = seg.filter_cloud(c, (0.3, -0.2, 0.1), -1.73, return_details=True)
        r = ref.filter_cloud(c, (0.3, -0.2, 0.1), -1.73)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), it
        assert out.tobytes() == r["out_points"].tobytes(), it
        if it % 10 == 9 or it in (2, 3, 41, 42, 43):
            for name in oracle.LAYERS:
                assert nan_equal(seg.map(0)[name], ref.layer(name)), (it, name)
    replays = seg.debug_set_tuning("graph_replays", 0) - before
    assert replays >= 80, replays  # (the first two calls of a kind are eager and capturing; five calls ran with graphs off)
    seg.close()


def test_results_straight_into_host_memory_or_copied_and_input_in_pieces():
    """The host calls let k_label write counts, index and labels into the pinned result block (results_direct, the default) or into HBM
    with a copy behind the kernel; the input travels in one piece (the default) or several.  Every combination, synchronous calls and
    the two-deep pipeline, ragged sizes and an empty cloud: the same bytes as the oracle."""
    clouds = [synth.hdl64_cloud(seed=90 + k, n_az=140 + 61 * k) for k in range(4)] + [synth.empty_cloud(0), synth.hdl64_cloud(seed=95, n_az=33)[:5]]
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=max(len(c) for c in clouds) + 64)
    ref = oracle.OracleMap(120.0, 0.33)
    it = 0
    for direct in (1, 0, 1):
        for pieces in (1, 3, 8):
            seg.debug_set_tuning("results_direct", direct)
            seg.debug_set_tuning("upload_pieces", pieces)
            for _ in range(3):
                c = clouds[it % len(clouds)]
                it += 1
                out, labels, index = seg.filter_cloud(c, (0.1, 0.2, 0.0), -1.73, return_details=True)
                r = ref.filter_cloud(c, (0.1, 0.2, 0.0), -1.73)
                assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), (direct, pieces, it)
                assert out.tobytes() == r["out_points"].tobytes(), (direct, pieces, it)
            # two deep: the second ticket is issued before the first is waited for
            a, b = clouds[it % len(clouds)], clouds[(it + 1) % len(clouds)]
            it += 2
            ta = seg.filter_cloud_async(a, (0.1, 0.2, 0.0), -1.73)
            tb = seg.filter_cloud_async(b, (0.1, 0.2, 0.0), -1.73)
            for t, c in ((ta, a), (tb, b)):
                out = seg.filter_cloud_wait(t)
                r = ref.filter_cloud(c, (0.1, 0.2, 0.0), -1.73)
                assert out.tobytes() == r["out_points"].tobytes(), (direct, pieces, "async")
    for name in oracle.LAYERS:
        assert nan_equal(seg.map(0)[name], ref.layer(name)), name
    seg.close()


@pytest.mark.parametrize("registered", [False, True])
def test_filter_cloud_with_layers_fused_call(registered):
    """gg_filter_cloud_layers: the returned cloud and all eleven layers of one call; the early layers travel while the sweep runs.
    Registered planes are written by the device directly, others through the staging block; both over several clouds (graph replay),
    with a subset of layers, and next to plain gg_filter_cloud / gg_get_layers calls on the same context."""
    clouds = [synth.hdl64_cloud(seed=80 + k, n_az=260 + 40 * k) for k in range(3)]
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=max(len(c) for c in clouds))
    ref = oracle.OracleMap(120.0, 0.33)
    planes = seg.alloc_layers(register=registered)
    some = {k: planes[k] for k in ("ground", "variance", "points")}
    for it in range(12):
        c = clouds[it % 3]
        use = some if it in (5, 6, 7) else planes
        if it == 9:  # a plain call in between: the next fused call sees its state
            seg.filter_cloud(c, ORIGIN0, -1.73)
            ref.filter_cloud(c, ORIGIN0, -1.73)
            continue
        for v in use.values():
            v[...] = np.float32(-7.0)
        out, labels, index = seg.filter_cloud_with_layers(c, ORIGIN0, -1.73, use, return_details=True)
        r = ref.filter_cloud(c, ORIGIN0, -1.73)
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]) and out.tobytes() == r["out_points"].tobytes(), it
        for name, v in use.items():
            assert nan_equal(v, ref.layer(name)), (it, name)
        if it == 3:
            for name in oracle.LAYERS:
                assert nan_equal(seg.map(0)[name], ref.layer(name)), name
    if registered:
        seg.release_layers(planes)
    seg.close()


def test_fused_call_with_lazily_materialised_layers_falls_back():
    cloud = synth.hdl64_cloud(seed=90, n_az=300)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    seg.set_flags(minimal_layers=True)
    ref = oracle.OracleMap(120.0, 0.33)
    planes = seg.alloc_layers(register=False)
    for it in range(3):
        seg.filter_cloud_with_layers(cloud, ORIGIN0, -1.73, planes)
        ref.filter_cloud(cloud, ORIGIN0, -1.73)
        for name, v in planes.items():
            assert nan_equal(v, ref.layer(name)), (it, name)
    seg.close()


def test_minimal_layers_reads_after_set_config_and_move():
    """the layers GG_FLAG_MINIMAL_LAYERS owes are computed from what the LAST cloud left in the slot -- also when the
    configuration has changed since (they are that cloud's: its classes were decided under the old configuration), and over a drive
    with the map scrolling between the clouds."""
    cloud = synth.hdl64_cloud(seed=91, n_az=300)
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=len(cloud))
    seg.set_flags(minimal_layers=True)
    ref = oracle.OracleMap(120.0, 0.33)
    owed = ("maxGroundHeight", "groundCandidates", "planeDist")
    seg.filter_cloud(cloud, ORIGIN0, -1.73)
    ref.filter_cloud(cloud, ORIGIN0, -1.73)
    cfg = seg.getConfig()
    cfg.max_ring = 40
    seg.setConfig(cfg)  # (blocking; rebuilds the patch table)
    for name in owed + ("minGroundHeight", "m2", "ground"):
        assert nan_equal(seg.map(0)[name], ref.layer(name)), name
    ref.cfg.max_ring = 40
    pose = (-0.7, 0.0, 1.73, 0.0, 0.0, 0.0, 1.0)
    for it in range(3):
        moved_gpu = seg.map(0).move(0.7 * (it + 1), 0.0, pose)
        moved_ref = ref.update(0.7 * (it + 1), 0.0, pose)
        assert moved_ref[1] == tuple(moved_gpu)
        seg.filter_cloud(cloud, (0.7 * (it + 1), 0.0, 0.0), -1.73)
        ref.filter_cloud(cloud, (0.7 * (it + 1), 0.0, 0.0), -1.73)
        for name in owed if it < 2 else oracle.LAYERS:
            assert nan_equal(seg.map(0)[name], ref.layer(name)), (it, name)
    seg.close()


# ---------------------------------------------------------------- GG_FLAG_CONCURRENT_HALVES

def _halves_inputs(B, stride, clouds):
    import torch

    host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
    for b, c in enumerate(clouds):
        host[b, : len(c)] = api.pack16(c)
    return torch.from_numpy(host.view(np.uint8).reshape(B, stride, 16)).cuda()


@pytest.mark.parametrize("n_slots,B", [(8, 8), (7, 5)])
def test_batches_as_two_concurrent_halves_match_the_oracle(n_slots, B):
    """The clouds of a batch whose maps are in the upper half of the context's slots run on the library's side stream, the others on
    the caller's; nothing joins them between steps.  Cold and warm steps with the clouds rotating over the slots, map
    re-initialisation on the caller's stream (divided the same way), getters, a small undivided batch and a profiled (hence
    undivided) batch in between: every output against the oracle."""
    import torch

    clouds = [synth.hdl64_cloud(seed=300 + k, n_az=120 + 31 * k) for k in range(B)]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=n_slots, max_points=stride)
    seg.set_flags(concurrent_halves=True)
    seg.debug_set_tuning("halves_min_clouds", 2)
    refs = [oracle.OracleMap(120.0, 0.33) for _ in range(n_slots)]
    pts = _halves_inputs(B, stride, clouds)
    n, org, bz = [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73)
    side = torch.cuda.Stream()
    out = None
    with torch.cuda.stream(side):
        for step in range(9):
            slots = [(b + 3 * step) % n_slots for b in range(B)]
            cold = step in (0, 4, 5)
            if cold:
                seg.reset_maps(0, n_slots, persistent_only=(step != 0), on_torch_stream=True)
                for r in refs:
                    if step != 0:  # (persistent_only: ground / groundpatch; the per-call layers are rewritten by the cloud anyway)
                        r.set_layer("ground", np.zeros((r.rows, r.cols), np.float32))
                        r.set_layer("groundpatch", np.full((r.rows, r.cols), np.float32(0.0000001)))
                    else:
                        r.reset_state()
            if step == 6:  # profiling switches the division off for this call: one launch sequence on the caller's stream
                seg.set_flags(concurrent_halves=True, profile=True)
            if step == 7:
                seg.set_flags(concurrent_halves=True)
            out = seg.filter_batch(pts, n, org, bz, out=out, slots=np.asarray(slots, np.int32))
            seg.batch_fence()  # the copies below run on this torch stream: after BOTH halves
            labels = out.labels.cpu().numpy()
            index = out.out_index.cpu().numpy()
            counts = out.counts.cpu().numpy()
            for b, c in enumerate(clouds):
                r = refs[slots[b]].filter_cloud(c, ORIGIN0, -1.73)
                assert np.array_equal(labels[b, : len(c)], r["label"]), (step, b)
                assert np.array_equal(index[b, : len(c)], r["index"]), (step, b)
                assert counts[b, 0] == len(r["out_points"]), (step, b)
            if step in (2, 8):  # getters order themselves after both halves
                for s in (0, n_slots - 1, n_slots // 2):
                    for name in ("ground", "groundpatch", "variance", "points"):
                        assert nan_equal(seg.map(s)[name], refs[s].layer(name)), (step, s, name)
            if step == 3:  # a batch too small to divide, on upper slots: it follows the second half of the batch before
                one = seg.filter_batch(pts[:1].contiguous(), n[:1], org[:1], bz[:1], slots=np.asarray([n_slots - 1], np.int32))
                seg.batch_fence()
                r = refs[n_slots - 1].filter_cloud(clouds[0], ORIGIN0, -1.73)
                assert np.array_equal(one.labels.cpu().numpy()[0, : n[0]], r["label"])
    seg.synchronize()
    seg.close()


@pytest.mark.parametrize("n_slots,B,geometry", [(8, 8, (120.0, 0.33)), (9, 6, (120.0, 0.33)), (6, 6, (200.0, 0.2))])
def test_unfenced_divided_batches_whose_halves_change(n_slots, B, geometry):
    """Several divided batches in a row on one stream WITHOUT a fence between them, the clouds rotating over the slots so that the halves
    change size and a row of the (reused) output buffers changes its half from step to step: the library has to order the two streams
    itself where a row would otherwise be written from both (enqueue_batch: output ranges x row-to-half map of the last divided
    batches), and the side stream's scan must not share its words with the caller stream's (a 1000 x 1000 map scans in parts).  Only the
    LAST step's outputs are read -- after one fence -- and every map against the oracle."""
    import torch

    length, resolution = geometry
    k = np.float32(length / 120.0)
    clouds = []
    for b in range(B):
        c = synth.hdl64_cloud(seed=500 + b, n_az=90 + 37 * b)
        c["x"] *= k
        c["y"] *= k
        clouds.append(c)
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(length, resolution, n_slots=n_slots, max_points=stride)
    seg.set_flags(concurrent_halves=True)
    seg.debug_set_tuning("halves_min_clouds", 2)
    refs = [oracle.OracleMap(length, resolution) for _ in range(n_slots)]
    pts = _halves_inputs(B, stride, clouds)
    n, org, bz = [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73)
    side = torch.cuda.Stream()
    outs = [None, None]
    steps = 7
    with torch.cuda.stream(side):
        for step in range(steps):
            slots = [(b * 2 + 3 * step) % n_slots if n_slots % 2 else (b + 3 * step) % n_slots for b in range(B)]
            if len(set(slots)) < B:  # (keep the slots of a batch distinct)
                slots = [(b + step) % n_slots for b in range(B)]
            outs[step % 2] = seg.filter_batch(pts, n, org, bz, out=outs[step % 2], slots=np.asarray(slots, np.int32))
            last = [refs[slots[b]].filter_cloud(c, ORIGIN0, -1.73) for b, c in enumerate(clouds)]
        seg.batch_fence()
        out = outs[(steps - 1) % 2]
        labels, index, counts = out.labels.cpu().numpy(), out.out_index.cpu().numpy(), out.counts.cpu().numpy()
    for b, c in enumerate(clouds):
        assert np.array_equal(labels[b, : len(c)], last[b]["label"]), b
        assert np.array_equal(index[b, : len(c)], last[b]["index"]), b
        assert counts[b, 0] == len(last[b]["out_points"]), b
    for s in range(n_slots):
        for name in ("ground", "groundpatch", "points"):
            assert nan_equal(seg.map(s)[name], refs[s].layer(name)), (s, name)
    seg.synchronize()
    seg.close()


def test_new_entry_points_on_empty_and_all_outside_clouds():
    """The reference's edge cases (empty cloud, every point outside the map) through the round-5 entry points: the fused call, the
    PointCloud2-out call and the sensor-frame transform inside the fused call."""
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=4096)
    ref = oracle.OracleMap(120.0, 0.33)
    planes = seg.alloc_layers(register=False)
    far = synth.make_cloud(np.column_stack([np.full(300, 500.0), np.linspace(-5, 5, 300), np.full(300, -1.7)]).astype(np.float32), ring=np.arange(300) % 64)
    near = synth.hdl64_cloud(seed=17, n_az=60)[:4000]
    tf = api.transform_from_pose((0.4, -0.2, 0.1, 0.0, 0.0, np.sin(-0.3), np.cos(-0.3)))
    for it, c in enumerate((synth.empty_cloud(0), far, near, synth.empty_cloud(0), near)):
        use_tf = it == 4
        ref_in = c
        if use_tf:
            ref_in = synth.clone_cloud(c)
            R, t = tf[:, :3], tf[:, 3]
            p = np.stack([c["x"], c["y"], c["z"]], 1).astype(np.float64)
            for k, name in enumerate(("x", "y", "z")):
                ref_in[name] = (((R[k, 0] * p[:, 0] + R[k, 1] * p[:, 1]) + R[k, 2] * p[:, 2]) + t[k]).astype(np.float32)
        out, labels, index = seg.filter_cloud_with_layers(c, ORIGIN0, -1.73, planes, return_details=True, map_from_cloud=tf if use_tf else None)
        r = ref.filter_cloud(ref_in, ORIGIN0, -1.73)
        assert len(out) == len(r["out_points"]) and out.tobytes() == r["out_points"].tobytes(), it
        assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"]), it
        for name, v in planes.items():
            assert nan_equal(v, ref.layer(name)), (it, name)
        wire = api.to_pc2(c)
        rec = seg.filter_cloud_pc2_out(wire.tobytes(), len(c), 18, (0, 4, 8, 16), ORIGIN0, -1.73)
        r2 = ref.filter_cloud(c, ORIGIN0, -1.73)
        assert rec.tobytes() == expected_pc2(r2["out_points"]), it
    seg.close()


def test_pc2_records_and_fused_layers_at_full_size():
    """BASELINE configs[1] and configs[3] at full size through the round-5 entry points: 125 k points / 364 x 364 and 2.1 M points /
    1000 x 1000 -- the 18-byte records of every returned point (both alignments of a record, all three parts of the returned cloud on
    the warm second frame of the small case) and all eleven layers of the fused call."""
    for cloud, length, resolution, frames in ((synth.hdl64_cloud(), 120.0, 0.33, 2), (synth.os128_cloud(seed=2), 200.0, 0.2, 1)):
        seg = api.GroundSegmentation().init(length, resolution, n_slots=1, max_points=len(cloud))
        ref = oracle.OracleMap(length, resolution)
        planes = seg.alloc_layers(register=True)
        wire = api.to_pc2(cloud).tobytes()
        for f in range(frames):
            rec = seg.filter_cloud_pc2_out(wire, len(cloud), 18, (0, 4, 8, 16), ORIGIN0, -1.73)
            r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
            assert rec.tobytes() == expected_pc2(r["out_points"]), (length, f)
            out = seg.filter_cloud_with_layers(cloud, ORIGIN0, -1.73, planes)
            r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
            assert out.tobytes() == r["out_points"].tobytes(), (length, f)
            for name, v in planes.items():
                assert nan_equal(v, ref.layer(name)), (length, f, name)
        seg.release_layers(planes)
        seg.close()


# ---------------------------------------------------------------- GroundSegmentation::insert_cloud as a member of its own (gg_insert_cloud)

@pytest.mark.parametrize("length,resolution", [(120.0, 0.33), (33.0, 0.33), (200.0, 0.2)])
def test_insert_cloud_in_ranges_continues_the_layers_as_they_stand(length, resolution):
    """GroundSegmentation::insert_cloud (include/groundgrid/GroundSegmentation.h:55, src/GroundSegmentation.cpp:200-311) called the way the
    header allows: sub-ranges of a cloud, one after the other, INTO a map that filter_cloud and earlier ranges already wrote -- no
    per-call reset, a cell's count continues from c > 0, the outlier test reads the terrain as it stands.  Per range the classes and
    cells (= the reference's three lists) and after every range all eleven layers against the oracle's insert_cloud on the same state."""
    k = np.float32(length / 120.0)
    cloud = synth.hdl64_cloud(seed=83, n_az=700)
    cloud["x"] *= k
    cloud["y"] *= k
    low = synth.clone_cloud(cloud)
    low["z"][::3] -= np.float32(0.9)                      # a third of the returns dive under the terrain: outliers once the map is warm
    seg = api.GroundSegmentation().init(length, resolution, n_slots=2, max_points=len(cloud))
    ref = oracle.OracleMap(length, resolution)
    m = seg.map(1)
    for _ in range(2):                                    # a warm map first (slot 1: the stage entry takes any slot)
        seg.filter_cloud(cloud, ORIGIN0, -1.73, map=m)
        ref.filter_cloud(cloud, ORIGIN0, -1.73)
    n = len(low)
    origin = (0.5, -0.25, 0.1)
    for start, end in ((0, n // 3), (n // 3, n // 3), (n // 3, n - 5), (n - 5, n), (0, n)):   # (an empty range; the whole cloud once more on top)
        cls, cell = m.insert_cloud(low, start, end, origin)
        rcls, rcell = ref.stage_insert(low[start:end], origin)
        assert np.array_equal(cls, rcls), (start, end)
        inside = rcls != oracle.OUTSIDE
        assert np.array_equal(cell[inside], rcell[inside]), (start, end)
        for name in oracle.LAYERS:
            assert nan_equal(m.get(name), ref.layer(name)), (start, end, name)
    assert (rcls == oracle.OUTLIER).sum() > 0 and ref.layer("points").max() > 3
    # a layer the host put there (counts that are not integers): the reference's expressions as they stand
    pts = ref.layer("points").copy()
    pts[pts > 0] += np.float32(0.25)
    m.set("points", pts)
    ref.set_layer("points", pts)
    cls, cell = m.insert_cloud(cloud, 0, n // 2, ORIGIN0)
    rcls, rcell = ref.stage_insert(cloud[: n // 2], ORIGIN0)
    assert np.array_equal(cls, rcls)
    for name in oracle.LAYERS:
        assert nan_equal(m.get(name), ref.layer(name)), name
    # and filter_cloud afterwards starts from its own reset as always
    out, labels, index = seg.filter_cloud(cloud, ORIGIN0, -1.73, map=m, return_details=True)
    r = ref.filter_cloud(cloud, ORIGIN0, -1.73)
    assert np.array_equal(labels, r["label"]) and np.array_equal(index, r["index"])
    for name in oracle.LAYERS:
        assert nan_equal(m.get(name), ref.layer(name)), name
    seg.close()
