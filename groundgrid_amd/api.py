"""Host-side mirror of the reference's operator interface for the hot path, on top of the C ABI.

``GroundSegmentation`` keeps the names and argument meaning of ``groundgrid::GroundSegmentation``
(/root/reference/include/groundgrid/GroundSegmentation.h:48-71): ``init``, ``setConfig``,
``filter_cloud``.  The grid map the reference borrows by reference (``grid_map::GridMap&``, owned by
``GroundGrid``) lives in HBM inside the context; ``GridMap`` is a handle to one such map ("slot") with
grid_map-like accessors.  ``filter_batch`` is the device-resident batched form (independent
(cloud, map) pairs in one set of launches) used for throughput runs and multi-GPU sharding.

PyTorch is used only for device buffers / streams in the batched path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import GGBatch, GGConfig, GGGeometry, GroundGridError, LAYERS
from .synth import POINT_DTYPE

# label / class codes (include/groundgrid_hip.h)
DROPPED, GROUND, NONGROUND = 0, 49, 99
OUTSIDE, IGNORED, OUTLIER, KEPT = 0, 1, 2, 3

POINT16_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("ring", "<u2"), ("pad", "<u2")])
# the 18-byte sensor_msgs/PointCloud2 record of scripts/kitti_data_publisher.py:139-150
PC2_DTYPE = np.dtype({"names": ["x", "y", "z", "intensity", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"], "offsets": [0, 4, 8, 12, 16], "itemsize": 18})


def to_pc2(cloud: np.ndarray) -> np.ndarray:
    """PointXYZIR (32 B) -> the 18-byte PointCloud2 records a publisher would send."""
    out = np.zeros(cloud.shape[0], dtype=PC2_DTYPE)
    for k in ("x", "y", "z", "intensity", "ring"):
        out[k] = cloud[k]
    return out


def default_config() -> GGConfig:
    c = GGConfig()
    _lib.load().gg_default_config(C.byref(c))
    return c


def transform_from_pose(pose7, rotation: str = "kdl") -> np.ndarray:
    """(tx, ty, tz, qx, qy, qz, qw) -> 3x4 (R | t) with the rotation built the way tf2::Matrix3x3::setRotation ("tf2") or
    KDL::Rotation::Quaternion ("kdl": what doTransform(PointStamped) goes through in ROS Noetic) builds it."""
    p = (C.c_double * 7)(*[float(v) for v in pose7])
    out = (C.c_double * 12)()
    rc = _lib.load().gg_transform_from_pose(_lib.ROTATION[rotation], p, out)
    if rc != _lib.GG_OK:
        raise GroundGridError(f"gg_transform_from_pose: {_lib.STATUS.get(rc, rc)}")
    return np.array(list(out), dtype=np.float64).reshape(3, 4)


def pack16(cloud: np.ndarray) -> np.ndarray:
    """PointXYZIR (32 B) -> packed 16-B device records (x, y, z, ring)."""
    out = np.zeros(cloud.shape[0], dtype=POINT16_DTYPE)
    out["x"], out["y"], out["z"], out["ring"] = cloud["x"], cloud["y"], cloud["z"], cloud["ring"]
    return out


def _check(L, ctx, rc, what):
    if rc != _lib.GG_OK:
        msg = L.gg_last_error(ctx).decode() if ctx else ""
        raise GroundGridError(f"{what}: {_lib.STATUS.get(rc, rc)} {msg}")


class GridMap:
    """Handle to one device-resident map state (the reference's grid_map::GridMap with its 11 layers)."""

    def __init__(self, seg: "GroundSegmentation", slot: int):
        self._seg = seg
        self.slot = slot
        self._pos = (0.0, 0.0)

    # grid_map-like accessors
    def getSize(self):
        return self._seg.rows, self._seg.cols

    def getResolution(self) -> float:
        return self._seg.resolution

    def getLength(self):
        return self._seg.length

    def getPosition(self):
        return self._pos

    def setPosition(self, x: float, y: float):
        """Map position after grid_map::move (src/GroundGrid.cpp:97)."""
        L, ctx = self._seg._L, self._seg._ctx
        _check(L, ctx, L.gg_set_map_position(ctx, self.slot, float(x), float(y)), "gg_set_map_position")
        self._pos = (float(x), float(y))

    def move(self, odom_x: float, odom_y: float, base_to_map=(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0), rotation: str = "kdl"):
        """GroundGrid::update (src/GroundGrid.cpp:83-147) on the device.  base_to_map = (tx, ty, tz, qx, qy, qz, qw) of
        lookupTransform("base_link", "map"); `rotation` picks the quaternion -> matrix convention of doTransform (the ABI
        itself takes the matrix entries).  Returns the index shift (rows, cols)."""
        L, ctx = self._seg._L, self._seg._ctx
        M = transform_from_pose(base_to_map, rotation)
        plane = (C.c_double * 4)(M[2, 0], M[2, 1], M[2, 2], M[2, 3])
        sh = (C.c_int * 2)()
        _check(L, ctx, L.gg_move_map(ctx, self.slot, float(odom_x), float(odom_y), plane, sh), "gg_move_map")
        x, y = C.c_double(), C.c_double()
        L.gg_get_map_position(ctx, self.slot, C.byref(x), C.byref(y))
        self._pos = (x.value, y.value)
        return sh[0], sh[1]

    def reset(self, odom_z: float = 0.0, pos=(0.0, 0.0)):
        """GroundGrid::initGroundGrid layer values (src/GroundGrid.cpp:71-75)."""
        L, ctx = self._seg._L, self._seg._ctx
        _check(L, ctx, L.gg_reset_map(ctx, self.slot, float(pos[0]), float(pos[1]), C.c_float(odom_z)), "gg_reset_map")
        self._pos = (float(pos[0]), float(pos[1]))

    def get(self, layer: str) -> np.ndarray:
        """Layer as a (rows, cols) float32 array (element (i, j) == Eigen's matrix(i, j))."""
        L, ctx = self._seg._L, self._seg._ctx
        buf = np.empty(self._seg.rows * self._seg.cols, dtype=np.float32)
        _check(L, ctx, L.gg_get_layer(ctx, self.slot, LAYERS.index(layer), buf.ctypes.data), "gg_get_layer")
        return buf.reshape((self._seg.rows, self._seg.cols), order="F")

    __getitem__ = get

    def set(self, layer: str, arr: np.ndarray):
        L, ctx = self._seg._L, self._seg._ctx
        a = np.asfortranarray(np.asarray(arr, dtype=np.float32))
        assert a.shape == (self._seg.rows, self._seg.cols)
        flat = np.ascontiguousarray(a.reshape(-1, order="F"))
        _check(L, ctx, L.gg_set_layer(ctx, self.slot, LAYERS.index(layer), flat.ctypes.data), "gg_set_layer")

    def layers(self, names=None) -> dict:
        """All (or the named) layers with one synchronisation (gg_get_layers): what a publisher loop reads after a cloud."""
        L, ctx = self._seg._L, self._seg._ctx
        names = list(LAYERS) if names is None else list(names)
        n = self._seg.rows * self._seg.cols
        bufs = {k: np.empty(n, dtype=np.float32) for k in names}
        ptrs = (C.c_void_p * len(LAYERS))(*[bufs[k].ctypes.data if k in bufs else None for k in LAYERS])
        _check(L, ctx, L.gg_get_layers(ctx, self.slot, ptrs), "gg_get_layers")
        return {k: v.reshape((self._seg.rows, self._seg.cols), order="F") for k, v in bufs.items()}

    def image_u8(self, layer: str):
        """GridMapCvConverter::toImage<unsigned char,1> (Nodelet.cpp:239): (rows x cols uint8 image, lower, upper)."""
        L, ctx = self._seg._L, self._seg._ctx
        img = np.empty((self._seg.rows, self._seg.cols), dtype=np.uint8)
        lo, hi = C.c_float(), C.c_float()
        _check(L, ctx, L.gg_get_layer_image_u8(ctx, self.slot, LAYERS.index(layer), img.ctypes.data, C.byref(lo), C.byref(hi)), "gg_get_layer_image_u8")
        return img, lo.value, hi.value

    def gridmap_message(self, layers=None, seq: int = 0, stamp=(0, 0), frame_id: str = "map", basic_layers=()) -> bytes:
        """The serialised grid_map_msgs/GridMap the nodelet publishes per cloud (Nodelet.cpp:211-214), ROS 1 wire format."""
        L, ctx = self._seg._L, self._seg._ctx
        mask = 0 if layers is None else sum(1 << LAYERS.index(k) for k in layers)
        hdr = _lib.GGGridMapHeader(int(seq), int(stamp[0]), int(stamp[1]), frame_id.encode(), sum(1 << LAYERS.index(k) for k in basic_layers))
        size = C.c_size_t(0)
        _check(L, ctx, L.gg_get_gridmap_message(ctx, self.slot, mask, C.byref(hdr), None, 0, C.byref(size)), "gg_get_gridmap_message")
        buf = np.empty(size.value, dtype=np.uint8)
        _check(L, ctx, L.gg_get_gridmap_message(ctx, self.slot, mask, C.byref(hdr), buf.ctypes.data, buf.size, C.byref(size)), "gg_get_gridmap_message")
        return buf.tobytes()

    # -- the stage members of the reference's class on this map as it stands (include/groundgrid/GroundSegmentation.h:59-62)
    def _stage(self, stage, section=0, i=0, j=0, base_z=0.0):
        L, ctx = self._seg._L, self._seg._ctx
        args = _lib.GGStageArgs(int(section), int(i), int(j), float(base_z))
        _check(L, ctx, L.gg_run_stage(ctx, self.slot, stage, C.byref(args)), "gg_run_stage")

    def detect_ground_patches(self, section: int = -1):
        """detect_ground_patches(map, section) (:314-340): section 0..3, or -1 for all four quadrants."""
        self._stage(_lib.GG_STAGE_DETECT_GROUND_PATCHES, section=section)

    def detect_ground_patch(self, S: int, i: int, j: int):
        assert S in (3, 5)
        self._stage(_lib.GG_STAGE_DETECT_GROUND_PATCH_3 if S == 3 else _lib.GG_STAGE_DETECT_GROUND_PATCH_5, i=i, j=j)

    def spiral_ground_interpolation(self, toBase_z: float):
        self._stage(_lib.GG_STAGE_SPIRAL_GROUND_INTERPOLATION, base_z=toBase_z)

    def interpolate_cell(self, x: int, y: int):
        self._stage(_lib.GG_STAGE_INTERPOLATE_CELL, i=x, j=y)

    def insert_cloud(self, cloud: np.ndarray, start: int, end: int, cloudOrigin: Sequence[float]):
        """GroundSegmentation::insert_cloud(cloud, start, end, cloudOrigin, point_index, ignored, outliers, map)
        (include/groundgrid/GroundSegmentation.h:55, src/GroundSegmentation.cpp:200-311) on this map as it stands (no per-call reset).
        Returns (class, cell) per point of [start, end) in cloud order; the three lists the reference appends to are the points of class
        KEPT / IGNORED (with their cells) / OUTLIER in that order."""
        L, ctx = self._seg._L, self._seg._ctx
        cloud = np.ascontiguousarray(cloud)
        assert cloud.dtype.itemsize == 32 and 0 <= start <= end <= len(cloud)
        n = end - start
        cls, cell = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.int32)
        org = (C.c_float * 3)(*[float(v) for v in cloudOrigin])
        _check(L, ctx, L.gg_insert_cloud(ctx, self.slot, cloud.ctypes.data, start, end, org, cls.ctypes.data, cell.ctypes.data), "gg_insert_cloud")
        return cls, cell

    def terrain_image(self) -> np.ndarray:
        """The 32FC3 terrain image of Nodelet.cpp:247-268: rows x cols x (ground, visited flag, pointsRaw)."""
        L, ctx = self._seg._L, self._seg._ctx
        img = np.empty((self._seg.rows, self._seg.cols, 3), dtype=np.float32)
        _check(L, ctx, L.gg_get_terrain_image(ctx, self.slot, img.ctypes.data), "gg_get_terrain_image")
        return img


@dataclass
class BatchOutputs:
    labels: "object"      # torch.uint8 [B, stride]
    out_index: "object"   # torch.int32 [B, stride]
    counts: "object"      # torch.int32 [B, 4]: returned size, kept, ignored, outliers
    out_clouds: "object" = None
    label_masks: "object" = None  # torch.uint8 [B, stride // 4]: 2 bits per point (0 dropped, 1 ground, 2 non-ground)
    out_pc2: "object" = None      # torch.uint8 [B, stride * 18]: the returned clouds as 18-byte PointCloud2 records


class GroundSegmentation:
    """Mirror of groundgrid::GroundSegmentation (include/groundgrid/GroundSegmentation.h:48-71)."""

    def __init__(self):
        self._L = _lib.load()
        self._ctx = None
        self._torch_used = False
        self._async = {}

    # -- GroundSegmentation::init(nodeHandle, dimension, resolution) (src/GroundSegmentation.cpp:37-48)
    def init(self, dimension: float = 120.0, resolution: float = 0.33, *, n_slots: int = 1,
             max_points: int = 150_000, device: int = 0, vertical_point_ang_dist: float = 0.0,
             min_dist_squared: float = 0.0):
        if self._ctx:
            self.close()
        geom = GGGeometry(float(dimension), float(resolution), float(vertical_point_ang_dist), float(min_dist_squared))
        ctx = C.c_void_p()
        rc = self._L.gg_create(C.byref(geom), int(n_slots), int(max_points), int(device), C.byref(ctx))
        if rc != _lib.GG_OK:
            raise GroundGridError(f"gg_create: {_lib.STATUS.get(rc, rc)}")
        self._ctx = ctx
        self.n_slots = n_slots
        self.max_points = max_points
        self.device = device
        r, c = C.c_int(), C.c_int()
        self._L.gg_get_size(ctx, C.byref(r), C.byref(c))
        self.rows, self.cols = r.value, c.value
        res, lx, ly = C.c_double(), C.c_double(), C.c_double()
        self._L.gg_get_geometry(ctx, C.byref(res), C.byref(lx), C.byref(ly))
        self.resolution, self.length = res.value, (lx.value, ly.value)
        self._maps = [GridMap(self, s) for s in range(n_slots)]
        return self

    def close(self):
        if self._ctx:
            self._L.gg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def map(self, slot: int = 0) -> GridMap:
        return self._maps[slot]

    def reset_maps(self, first_slot: int = 0, n_slots: Optional[int] = None, odom_z: float = 0.0, pos=(0.0, 0.0), persistent_only: bool = False,
                   on_torch_stream: bool = False):
        """GroundGrid::initGroundGrid values (src/GroundGrid.cpp:71-75) for a range of map states in one launch; with
        persistent_only just ground / groundpatch, the state that outlives a cloud (a "cold" start).  on_torch_stream: enqueue on
        the current torch stream (where filter_batch runs) instead of the context's own."""
        n = self.n_slots - first_slot if n_slots is None else n_slots
        stream = None
        if on_torch_stream:
            import torch

            h = torch.cuda.current_stream(self.device).cuda_stream
            stream = C.c_void_p(h) if h else C.c_void_p(-1)  # (0 = torch's default stream = GG_STREAM_DEFAULT)
        _check(self._L, self._ctx, self._L.gg_reset_maps(self._ctx, first_slot, n, float(pos[0]), float(pos[1]), C.c_float(odom_z),
                                                          1 if persistent_only else 0, stream), "gg_reset_maps")
        for s in range(first_slot, first_slot + n):
            self._maps[s]._pos = (float(pos[0]), float(pos[1]))

    # -- GroundSegmentation::setConfig (src/GroundSegmentation.cpp:468-471)
    def setConfig(self, config: GGConfig):
        _check(self._L, self._ctx, self._L.gg_set_config(self._ctx, C.byref(config)), "gg_set_config")

    def getConfig(self) -> GGConfig:
        c = GGConfig()
        _check(self._L, self._ctx, self._L.gg_get_config(self._ctx, C.byref(c)), "gg_get_config")
        return c

    def set_flags(self, minimal_layers: bool = False, profile: bool = False, concurrent_halves: bool = False, eager_layers: bool = False):
        """gg_set_flags.  minimal_layers: maxGroundHeight / groundCandidates / planeDist -- written by insert_cloud
        (src/GroundSegmentation.cpp:296,303,307) and read by nothing on the path -- are computed when a layer getter asks for one of
        them instead of for every cloud; every getter still returns what the reference's layer would hold.  profile: per-kernel
        events (kernel_times).  filter_batch (device-resident clouds) does that by default; eager_layers switches it back to all nine
        per-call layers per cloud."""
        f = ((_lib.GG_FLAG_MINIMAL_LAYERS if minimal_layers else 0) | (_lib.GG_FLAG_PROFILE if profile else 0) |
             (_lib.GG_FLAG_CONCURRENT_HALVES if concurrent_halves else 0) | (_lib.GG_FLAG_EAGER_LAYERS if eager_layers else 0))
        _check(self._L, self._ctx, self._L.gg_set_flags(self._ctx, f), "gg_set_flags")

    def expected_points(self) -> np.ndarray:
        buf = np.empty(self.rows * self.cols, dtype=np.float32)
        _check(self._L, self._ctx, self._L.gg_get_expected_points(self._ctx, buf.ctypes.data), "gg_get_expected_points")
        return buf.reshape((self.rows, self.cols), order="F")

    # -- GroundSegmentation::filter_cloud (include/groundgrid/GroundSegmentation.h:54)
    def _host_buffers(self, n: int, reuse: bool):
        """Output arrays of one host-buffer call: fresh ones, or (reuse=True) this object's own, grown on demand -- a caller that
        looks at one result before asking for the next (a sensor loop) then pays no allocation and no first-touch page faults
        per cloud (3.8 MB per HDL-64E revolution); the returned arrays are views that the next reuse=True call overwrites."""
        m = max(n, 1)
        if not reuse:
            return np.empty(m * 32, dtype=np.uint8).view(POINT_DTYPE), np.empty(m, dtype=np.uint8), np.empty(m, dtype=np.int32)
        if getattr(self, "_hb_cap", 0) < m:
            self._hb_cap = m + m // 8
            self._hb = (np.zeros(self._hb_cap * 32, dtype=np.uint8).view(POINT_DTYPE), np.zeros(self._hb_cap, dtype=np.uint8),
                        np.zeros(self._hb_cap, dtype=np.int32))
        return self._hb

    def filter_cloud(self, cloud: np.ndarray, cloudOrigin: Sequence[float], mapToBase_z: float, map: Optional[GridMap] = None,
                     return_details: bool = False, map_from_cloud=None, reuse_buffers: bool = False):
        """cloud: POINT_DTYPE array in the map frame.  Returns the segmented cloud (intensity = 49 ground /
        99 non-ground; order kept, ignored, outliers).  With return_details also (labels, out_index)."""
        assert cloud.dtype == POINT_DTYPE, "cloud must use groundgrid_amd.synth.POINT_DTYPE (PointXYZIR, 32 B)"
        gm = map if map is not None else self._maps[0]
        cloud = np.ascontiguousarray(cloud)
        n = cloud.shape[0]
        out, labels, index = self._host_buffers(n, reuse_buffers)
        out_n = C.c_size_t(0)
        org = (C.c_float * 3)(*[float(v) for v in cloudOrigin])
        # (the per-point labels / positions are copied out only when asked for: the reference's call returns the cloud alone)
        lab_p, idx_p = (labels.ctypes.data, index.ctypes.data) if return_details else (None, None)
        if map_from_cloud is None:
            rc = self._L.gg_filter_cloud(self._ctx, gm.slot, cloud.ctypes.data, n, org, float(mapToBase_z),
                                         out.ctypes.data, C.byref(out_n), lab_p, idx_p)
        else:  # cloud still in the sensor frame: 3x4 (R | t) of map <- cloud frame, transformed on the device
            tf = (C.c_double * 12)(*[float(v) for v in np.asarray(map_from_cloud, dtype=np.float64).reshape(-1)[:12]])
            rc = self._L.gg_filter_cloud_tf(self._ctx, gm.slot, cloud.ctypes.data, n, tf, org, float(mapToBase_z),
                                            out.ctypes.data, C.byref(out_n), lab_p, idx_p)
        _check(self._L, self._ctx, rc, "gg_filter_cloud")
        seg = out[: out_n.value]
        if return_details:
            return seg, labels[:n], index[:n]
        return seg

    segment = filter_cloud  # BASELINE.json's north_star calls the entry point segment(); same thing

    def alloc_layers(self, names=None, register: bool = True) -> dict:
        """Host planes for filter_cloud_with_layers: one (rows, cols) Fortran-ordered float32 array per layer (Eigen::MatrixXf storage),
        registered with the context (gg_host_register) so that the device writes them directly -- what a host does once with the
        planes of its grid_map::GridMap."""
        names = list(LAYERS) if names is None else list(names)
        planes = {k: np.zeros((self.rows, self.cols), dtype=np.float32, order="F") for k in names}
        if register:
            for v in planes.values():
                _check(self._L, self._ctx, self._L.gg_host_register(self._ctx, v.ctypes.data, v.nbytes), "gg_host_register")
        return planes

    def release_layers(self, planes: dict):
        for v in planes.values():
            self._L.gg_host_unregister(self._ctx, v.ctypes.data)

    def filter_cloud_with_layers(self, cloud: np.ndarray, cloudOrigin: Sequence[float], mapToBase_z: float, planes: dict, map: Optional[GridMap] = None,
                                 return_details: bool = False, map_from_cloud=None, reuse_buffers: bool = False):
        """filter_cloud and the download of the layers in `planes` ({name: (rows, cols) F-ordered float32 array}) as ONE call
        (gg_filter_cloud_layers): what the reference's nodelet does per cloud -- filter, then publish every layer
        (src/GroundGridNodelet.cpp:196-228) -- with the layer traffic overlapping the terrain sweep."""
        assert cloud.dtype == POINT_DTYPE
        gm = map if map is not None else self._maps[0]
        cloud = np.ascontiguousarray(cloud)
        n = cloud.shape[0]
        out, labels, index = self._host_buffers(n, reuse_buffers)
        out_n = C.c_size_t(0)
        org = (C.c_float * 3)(*[float(v) for v in cloudOrigin])
        lab_p, idx_p = (labels.ctypes.data, index.ctypes.data) if return_details else (None, None)
        tf = None
        if map_from_cloud is not None:
            tf = (C.c_double * 12)(*[float(v) for v in np.asarray(map_from_cloud, dtype=np.float64).reshape(-1)[:12]])
        for k, v in planes.items():
            assert v.dtype == np.float32 and v.shape == (self.rows, self.cols) and v.flags.f_contiguous, k
        ptrs = (C.c_void_p * len(LAYERS))(*[planes[k].ctypes.data if k in planes else None for k in LAYERS])
        rc = self._L.gg_filter_cloud_layers(self._ctx, gm.slot, cloud.ctypes.data, n, tf, org, float(mapToBase_z), out.ctypes.data, C.byref(out_n),
                                            lab_p, idx_p, ptrs)
        _check(self._L, self._ctx, rc, "gg_filter_cloud_layers")
        seg = out[: out_n.value]
        if return_details:
            return seg, labels[:n], index[:n]
        return seg

    def filter_cloud_pc2(self, data: bytes, n: int, point_step: int, offsets, cloudOrigin, mapToBase_z: float,
                         map: Optional[GridMap] = None, map_from_cloud=None):
        """filter_cloud straight from a sensor_msgs/PointCloud2 payload; offsets = (x, y, z, ring) byte offsets.
        Returns (labels, out_index, n_returned) per input point."""
        gm = map if map is not None else self._maps[0]
        buf = np.frombuffer(data, dtype=np.uint8)
        assert buf.size >= n * point_step
        labels = np.zeros(max(n, 1), dtype=np.uint8)
        index = np.zeros(max(n, 1), dtype=np.int32)
        out_n = C.c_size_t(0)
        org = (C.c_float * 3)(*[float(v) for v in cloudOrigin])
        tf = None
        if map_from_cloud is not None:
            tf = (C.c_double * 12)(*[float(v) for v in np.asarray(map_from_cloud, dtype=np.float64).reshape(-1)[:12]])
        rc = self._L.gg_filter_cloud_pc2(self._ctx, gm.slot, buf.ctypes.data, n, point_step, offsets[0], offsets[1], offsets[2], offsets[3],
                                         tf, org, float(mapToBase_z), labels.ctypes.data, index.ctypes.data, C.byref(out_n))
        _check(self._L, self._ctx, rc, "gg_filter_cloud_pc2")
        return labels[:n], index[:n], out_n.value

    def filter_cloud_pc2_out(self, data: bytes, n: int, point_step: int, offsets, cloudOrigin, mapToBase_z: float,
                             map: Optional[GridMap] = None, map_from_cloud=None) -> np.ndarray:
        """PointCloud2 payload in, PointCloud2 payload out: the returned cloud as 18-byte records (x, y, z, intensity, ring --
        PC2_DTYPE), written by the label kernel.  Returns the structured array of the returned points."""
        gm = map if map is not None else self._maps[0]
        buf = np.frombuffer(data, dtype=np.uint8)
        assert buf.size >= n * point_step
        out = np.empty(max(n, 1) * _lib.GG_PC2_POINT_STEP, dtype=np.uint8)
        out_n = C.c_size_t(0)
        org = (C.c_float * 3)(*[float(v) for v in cloudOrigin])
        tf = None
        if map_from_cloud is not None:
            tf = (C.c_double * 12)(*[float(v) for v in np.asarray(map_from_cloud, dtype=np.float64).reshape(-1)[:12]])
        rc = self._L.gg_filter_cloud_pc2_out(self._ctx, gm.slot, buf.ctypes.data, n, point_step, offsets[0], offsets[1], offsets[2], offsets[3],
                                             tf, org, float(mapToBase_z), out.ctypes.data, C.byref(out_n))
        _check(self._L, self._ctx, rc, "gg_filter_cloud_pc2_out")
        return out[: out_n.value * _lib.GG_PC2_POINT_STEP].view(PC2_DTYPE)

    # -- insert_cloud's per-point decision (include/groundgrid/GroundSegmentation.h:55)
    def point_classes(self, n: int, map: Optional[GridMap] = None):
        gm = map if map is not None else self._maps[0]
        cls = np.zeros(max(n, 1), dtype=np.uint8)
        cell = np.zeros(max(n, 1), dtype=np.int32)
        rc = self._L.gg_get_point_classes(self._ctx, gm.slot, n, cls.ctypes.data, cell.ctypes.data)
        _check(self._L, self._ctx, rc, "gg_get_point_classes")
        return cls[:n], cell[:n]

    # -- batched device-resident form
    def filter_batch(self, points, n_points: Sequence[int], origins, base_z, *, first_slot: int = 0,
                     out: Optional[BatchOutputs] = None, want_clouds: bool = False, want_masks: bool = False, stream=None,
                     transforms=None, slots=None, want_pc2: bool = False) -> BatchOutputs:
        """points: CUDA torch tensor [B, stride, 16] (packed gg_point16) or [B, stride, 32] (PointXYZIR), uint8.
        Enqueues on the current torch stream and returns without synchronising."""
        import torch

        self._torch_used = True
        assert points.is_cuda and points.dtype == torch.uint8 and points.dim() == 3 and points.is_contiguous()
        B, stride, rec = points.shape
        assert rec in (16, 32)
        fmt = _lib.GG_POINT16 if rec == 16 else _lib.GG_POINT32
        if out is None:
            out = BatchOutputs(
                labels=torch.empty((B, stride), dtype=torch.uint8, device=points.device),
                out_index=torch.empty((B, stride), dtype=torch.int32, device=points.device),
                counts=torch.empty((B, 4), dtype=torch.int32, device=points.device),
                out_clouds=torch.empty((B, stride, 32), dtype=torch.uint8, device=points.device) if want_clouds else None,
                label_masks=torch.zeros((B, stride // 4), dtype=torch.uint8, device=points.device) if want_masks else None,
                out_pc2=torch.empty((B, stride * _lib.GG_PC2_POINT_STEP), dtype=torch.uint8, device=points.device) if want_pc2 else None,
            )
        npts = (C.c_int32 * B)(*[int(v) for v in n_points])
        org = np.ascontiguousarray(np.asarray(origins, dtype=np.float32).reshape(B, 3))
        bz = np.ascontiguousarray(np.asarray(base_z, dtype=np.float64).reshape(B))
        b = GGBatch()
        b.n_clouds, b.first_slot, b.point_format = B, first_slot, fmt
        b.d_points, b.cloud_stride = points.data_ptr(), stride
        b.n_points = npts
        b.origins = org.ctypes.data_as(C.POINTER(C.c_float))
        b.base_z = bz.ctypes.data_as(C.POINTER(C.c_double))
        if transforms is not None:  # [B, 3, 4] map <- cloud frame: the per-point transform is fused into K1
            tfs = np.ascontiguousarray(np.asarray(transforms, dtype=np.float64).reshape(B, 12))
            b.transforms = tfs.ctypes.data_as(C.POINTER(C.c_double))
        if slots is not None:  # cloud b meets map slot slots[b] (distinct) instead of first_slot + b
            sl = (C.c_int32 * B)(*[int(v) for v in slots])
            b.slots = sl
        b.d_labels = out.labels.data_ptr()
        b.d_out_index = out.out_index.data_ptr()
        b.d_out_clouds = out.out_clouds.data_ptr() if out.out_clouds is not None else None
        b.d_out_counts = out.counts.data_ptr()
        b.d_label_masks = out.label_masks.data_ptr() if out.label_masks is not None else None
        b.d_out_pc2 = out.out_pc2.data_ptr() if out.out_pc2 is not None else None
        s = stream if stream is not None else torch.cuda.current_stream(points.device).cuda_stream
        # torch hands out 0 for its default stream = the legacy null stream; NULL would mean "the context's own stream" to
        # the library, which is not ordered with torch ops / RCCL -- so name the default stream explicitly
        rc = self._L.gg_filter_batch(self._ctx, C.byref(b), C.c_void_p(s if s else _lib.GG_STREAM_DEFAULT))
        _check(self._L, self._ctx, rc, "gg_filter_batch")
        return out

    def batch_fence(self, stream=None):
        """GG_FLAG_CONCURRENT_HALVES: order the current torch stream (or `stream`) after both halves of the batches enqueued so far --
        before anything the caller enqueues itself reads their outputs."""
        import torch

        h = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        _check(self._L, self._ctx, self._L.gg_batch_fence(self._ctx, C.c_void_p(h if h else _lib.GG_STREAM_DEFAULT)), "gg_batch_fence")

    def kernel_times(self, reset: bool = True):
        """(ms[7], launches[7]) accumulated under set_flags(profile=True)."""
        ms = (C.c_double * _lib.GG_NUM_KERNELS)()
        ln = (C.c_int64 * _lib.GG_NUM_KERNELS)()
        _check(self._L, self._ctx, self._L.gg_get_kernel_times(self._ctx, ms, ln, 1 if reset else 0), "gg_get_kernel_times")
        names = [self._L.gg_kernel_name(k).decode() for k in range(_lib.GG_NUM_KERNELS)]
        return {names[k]: (ms[k], ln[k]) for k in range(_lib.GG_NUM_KERNELS)}

    def synchronize(self):
        """Waits for everything the context has enqueued, batches on caller streams included (the library orders its own
        stream after them, include/groundgrid_hip.h gg_filter_batch)."""
        _check(self._L, self._ctx, self._L.gg_synchronize(self._ctx), "gg_synchronize")

    # -- pipelined reference-shaped call (gg_filter_cloud_async / gg_filter_cloud_wait)
    def filter_cloud_async(self, cloud: np.ndarray, cloudOrigin: Sequence[float], mapToBase_z: float, map: Optional[GridMap] = None,
                           map_from_cloud=None) -> int:
        """Enqueue one cloud and return a ticket; at most GG_ASYNC_DEPTH tickets may be outstanding."""
        assert cloud.dtype == POINT_DTYPE
        gm = map if map is not None else self._maps[0]
        cloud = np.ascontiguousarray(cloud)
        org = (C.c_float * 3)(*[float(v) for v in cloudOrigin])
        tf = None
        if map_from_cloud is not None:
            tf = (C.c_double * 12)(*[float(v) for v in np.asarray(map_from_cloud, dtype=np.float64).reshape(-1)[:12]])
        t = C.c_int(-1)
        rc = self._L.gg_filter_cloud_async(self._ctx, gm.slot, cloud.ctypes.data, cloud.shape[0], tf, org, float(mapToBase_z), C.byref(t))
        _check(self._L, self._ctx, rc, "gg_filter_cloud_async")
        self._async[t.value] = cloud  # keeps the input alive until the wait
        return t.value

    def filter_cloud_wait(self, ticket: int, return_details: bool = False, want_cloud: bool = True, reuse_buffers: bool = False):
        cloud = self._async.pop(ticket)
        n = cloud.shape[0]
        out, labels, index = self._host_buffers(n, reuse_buffers)
        if not want_cloud:
            out = None
        out_n = C.c_size_t(0)
        rc = self._L.gg_filter_cloud_wait(self._ctx, ticket, out.ctypes.data if want_cloud else None, C.byref(out_n),
                                          labels.ctypes.data if return_details else None, index.ctypes.data if return_details else None)
        _check(self._L, self._ctx, rc, "gg_filter_cloud_wait")
        seg = out[: out_n.value] if want_cloud else None
        if return_details:
            return seg, labels[:n], index[:n]
        return seg

    def debug_set_tuning(self, key: str, value: int) -> int:
        """Tools / tests: force a launch geometry the library would otherwise derive from the batch size ("sweep_waves",
        "k2_per_cloud", "k2_dense_share"; 0 = default); key "pw" returns the context's points per wave chunk."""
        fn = self._L.gg_debug_set_tuning
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        rc = fn(self._ctx, key.encode(), int(value))
        if rc < 0:
            raise GroundGridError(f"gg_debug_set_tuning({key}): {_lib.STATUS.get(rc, rc)}")
        return rc

    def set_conventions(self, eigen_reduction: int = 0):
        """Which Eigen the reference is built against (0 = 3.3.x order of the 5x5 block sums, 1 = 3.4.x SSE2)."""
        c = _lib.GGConventions()
        c.eigen_reduction = int(eigen_reduction)
        _check(self._L, self._ctx, self._L.gg_set_conventions(self._ctx, C.byref(c)), "gg_set_conventions")
