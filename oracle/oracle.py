"""ctypes binding of the CPU oracle (oracle/libgg_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (groundgrid_amd) never does.  PARITY UNPINNED -- see gg_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GG_ORACLE_LIB") or os.path.join(_HERE, "libgg_oracle.so")  # (override: the sanitizer build of tests/test_sanitizers_cpu.py)

# include/velodyne_pointcloud/point_types.h:27-33 -> 32-byte record
POINT_DTYPE = np.dtype(
    {
        "names": ["x", "y", "z", "intensity", "ring"],
        "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"],
        "offsets": [0, 4, 8, 16, 20],
        "itemsize": 32,
    }
)

LAYERS = [
    "points",
    "ground",
    "groundpatch",
    "minGroundHeight",
    "maxGroundHeight",
    "groundCandidates",
    "planeDist",
    "m2",
    "meanVariance",
    "pointsRaw",
    "variance",
]
NUM_LAYERS = len(LAYERS)

OUTSIDE, IGNORED, OUTLIER, KEPT = 0, 1, 2, 3
DROPPED, GROUND, NONGROUND = 0, 49, 99


class Config(C.Structure):
    """cfg/GroundGrid.cfg:8-21"""

    _fields_ = [
        ("point_count_cell_variance_threshold", C.c_int),
        ("max_ring", C.c_int),
        ("groundpatch_detection_minimum_threshold", C.c_double),
        ("distance_factor", C.c_double),
        ("minimum_distance_factor", C.c_double),
        ("miminum_point_height_threshold", C.c_double),
        ("minimum_point_height_obstacle_threshold", C.c_double),
        ("outlier_tolerance", C.c_double),
        ("ground_patch_detection_minimum_point_count_threshold", C.c_double),
        ("patch_size_change_distance", C.c_double),
        ("occupied_cells_decrease_factor", C.c_double),
        ("occupied_cells_point_count_factor", C.c_double),
        ("min_outlier_detection_ground_confidence", C.c_double),
        ("thread_count", C.c_int),
    ]


class _Map(C.Structure):
    _fields_ = [
        ("rows", C.c_int),
        ("cols", C.c_int),
        ("resolution", C.c_double),
        ("length", C.c_double * 2),
        ("position", C.c_double * 2),
        ("verticalPointAngDist", C.c_float),
        ("minDistSquared", C.c_float),
        ("layer", C.POINTER(C.c_float) * NUM_LAYERS),
        ("expectedPoints", C.POINTER(C.c_float)),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/gg_oracle.c -> libgg_oracle.so (gcc); returns the library path."""
    src = os.path.join(_HERE, "gg_oracle.c")
    hdr = os.path.join(_HERE, "gg_oracle.h")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgg_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ggo_default_config.argtypes = [C.POINTER(Config)]
        L.ggo_map_create.restype = C.POINTER(_Map)
        L.ggo_map_create.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double, C.c_float]
        L.ggo_map_destroy.argtypes = [C.POINTER(_Map)]
        L.ggo_map_reset_state.argtypes = [C.POINTER(_Map), C.c_double, C.c_double, C.c_float]
        L.ggo_filter_cloud.restype = C.c_size_t
        L.ggo_filter_cloud.argtypes = [
            C.POINTER(_Map), C.POINTER(Config), C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_double,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        L.ggo_filter_cloud_threads.restype = C.c_size_t
        L.ggo_filter_cloud_threads.argtypes = [C.POINTER(_Map), C.POINTER(Config), C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_double,
                                               C.c_int, C.c_void_p]
        L.ggo_stage_reset.argtypes = [C.POINTER(_Map)]
        L.ggo_stage_insert.argtypes = [
            C.POINTER(_Map), C.POINTER(Config), C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_void_p, C.c_void_p,
        ]
        L.ggo_stage_detect.argtypes = [C.POINTER(_Map), C.POINTER(Config)]
        L.ggo_stage_spiral.argtypes = [C.POINTER(_Map), C.POINTER(Config), C.c_double]
        L.ggo_stage_detect_section.argtypes = [C.POINTER(_Map), C.POINTER(Config), C.c_ushort]
        L.ggo_detect_ground_patch.argtypes = [C.POINTER(_Map), C.POINTER(Config), C.c_int, C.c_size_t, C.c_size_t]
        L.ggo_interpolate_cell.argtypes = [C.POINTER(_Map), C.POINTER(Config), C.c_size_t, C.c_size_t]
        L.ggo_set_eigen_reduction.argtypes = [C.c_int]
        L.ggo_get_eigen_reduction.restype = C.c_int
        L.ggo_rotation_from_quaternion.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ggo_rotation_from_quaternion.restype = None
        L.ggo_map_update.restype = C.c_int
        L.ggo_map_update.argtypes = [C.POINTER(_Map), C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ggo_get_index.restype = C.c_int
        L.ggo_get_index.argtypes = [C.POINTER(_Map), C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ggo_tree_sum.restype = C.c_float
        L.ggo_tree_sum.argtypes = [C.POINTER(C.c_float), C.c_int]
        L.ggo_block_sum.restype = C.c_float
        L.ggo_block_sum.argtypes = [C.POINTER(C.c_float), C.c_int]
        L.ggo_hypotf.restype = C.c_float
        L.ggo_hypotf.argtypes = [C.c_float, C.c_float]
        L.ggo_spiral_visit_count.restype = C.c_size_t
        L.ggo_spiral_visit_count.argtypes = [C.c_int]
        _lib = L
    return _lib


ROTATION = {"tf2": 0, "kdl": 1}


def rotation_from_quaternion(q_xyzw, rotation: str = "kdl") -> np.ndarray:
    """3x3 rotation the way tf2::Matrix3x3::setRotation ("tf2") or KDL::Rotation::Quaternion ("kdl") builds it."""
    q = (C.c_double * 4)(*[float(v) for v in q_xyzw])
    R = (C.c_double * 9)()
    lib().ggo_rotation_from_quaternion(ROTATION[rotation], q, R)
    return np.array(list(R), dtype=np.float64).reshape(3, 3)


def plane_from_pose(pose7, rotation: str = "kdl"):
    """(tx, ty, tz, qx, qy, qz, qw) -> (r20, r21, r22, tz): what GroundGrid::update needs of base_link <- map."""
    R = rotation_from_quaternion(pose7[3:7], rotation)
    return float(R[2, 0]), float(R[2, 1]), float(R[2, 2]), float(pose7[2])


def matrix_from_pose(pose7, rotation: str = "kdl") -> np.ndarray:
    """(tx, ty, tz, qx, qy, qz, qw) -> 3x4 (R | t)."""
    R = rotation_from_quaternion(pose7[3:7], rotation)
    return np.concatenate([R, np.asarray(pose7[:3], dtype=np.float64).reshape(3, 1)], axis=1)


def set_eigen_reduction(order: int):
    """0 = Eigen 3.3.x order of the 5x5 block sums (default), 1 = Eigen 3.4.x SSE2 (process-wide)."""
    lib().ggo_set_eigen_reduction(int(order))


def default_config() -> Config:
    c = Config()
    lib().ggo_default_config(C.byref(c))
    return c


def make_cloud(xyz: np.ndarray, ring=None, intensity=None) -> np.ndarray:
    n = xyz.shape[0]
    raw = np.zeros(n, dtype=np.uint8).repeat(32).reshape(n, 32)  # zeroed padding
    pts = raw.view(POINT_DTYPE).reshape(n)
    pts["x"] = xyz[:, 0]
    pts["y"] = xyz[:, 1]
    pts["z"] = xyz[:, 2]
    if ring is not None:
        pts["ring"] = ring
    if intensity is not None:
        pts["intensity"] = intensity
    return pts


class OracleMap:
    """One grid map + the reference algorithm state (ground / groundpatch persist across clouds)."""

    def __init__(self, length=120.0, resolution=0.33, pos=(0.0, 0.0), odom_z=0.0):
        self._L = lib()
        self._m = self._L.ggo_map_create(C.c_float(length), C.c_float(resolution), pos[0], pos[1], C.c_float(odom_z))
        if not self._m:
            raise ValueError("inconsistent geometry (grid_map size != GroundSegmentation::init cell count)")
        self.rows = self._m.contents.rows
        self.cols = self._m.contents.cols
        self.cfg = default_config()

    def __del__(self):
        try:
            if self._m:
                self._L.ggo_map_destroy(self._m)
                self._m = None
        except Exception:
            pass

    @property
    def resolution(self) -> float:
        return self._m.contents.resolution

    @property
    def length(self):
        return tuple(self._m.contents.length)

    @property
    def position(self):
        return tuple(self._m.contents.position)

    def reset_state(self, pos=(0.0, 0.0), odom_z=0.0):
        self._L.ggo_map_reset_state(self._m, pos[0], pos[1], C.c_float(odom_z))

    def update(self, odom_x: float, odom_y: float, base_to_map, rotation: str = "kdl"):
        """GroundGrid::update.  base_to_map = (tx, ty, tz, qx, qy, qz, qw) of lookupTransform("base_link", "map");
        `rotation` = which quaternion -> matrix convention doTransform(PointStamped) follows ("kdl" or "tf2").
        Returns (moved, (shift_rows, shift_cols))."""
        plane = (C.c_double * 4)(*plane_from_pose(base_to_map, rotation))
        sh = (C.c_int * 2)()
        moved = self._L.ggo_map_update(self._m, float(odom_x), float(odom_y), plane, sh)
        return bool(moved), (sh[0], sh[1])

    def filter_cloud_threads(self, cloud: np.ndarray, origin=(0.0, 0.0, 0.0), base_z=0.0, t_insert: int = 8):
        """TIMING ONLY: the reference's default threading shape (8 racing insertion threads + 4 detection threads, see
        gg_oracle.h): not deterministic, never a checker.  Returns the label array."""
        cloud = np.ascontiguousarray(cloud)
        n = cloud.shape[0]
        label = np.zeros(max(n, 1), dtype=np.uint8)
        org = (C.c_float * 3)(*[float(v) for v in origin])
        self._L.ggo_filter_cloud_threads(self._m, C.byref(self.cfg), cloud.ctypes.data, n, org, float(base_z), int(t_insert), label.ctypes.data)
        return label[:n]

    def layer(self, name: str) -> np.ndarray:
        """View (no copy) of a layer as (rows, cols) Fortran-ordered float32 (Eigen column-major)."""
        idx = LAYERS.index(name)
        ptr = self._m.contents.layer[idx]
        flat = np.ctypeslib.as_array(ptr, shape=(self.rows * self.cols,))
        return flat.reshape((self.rows, self.cols), order="F")

    def expected_points(self) -> np.ndarray:
        flat = np.ctypeslib.as_array(self._m.contents.expectedPoints, shape=(self.rows * self.cols,))
        return flat.reshape((self.rows, self.cols), order="F")

    def layers_copy(self) -> dict:
        return {n: self.layer(n).copy() for n in LAYERS}

    def set_layer(self, name: str, arr: np.ndarray):
        self.layer(name)[...] = np.asarray(arr, dtype=np.float32)

    def get_index(self, x: float, y: float):
        r, c = C.c_int(), C.c_int()
        inside = self._L.ggo_get_index(self._m, x, y, C.byref(r), C.byref(c))
        return bool(inside), r.value, c.value

    def filter_cloud(self, cloud: np.ndarray, origin=(0.0, 0.0, 0.0), base_z=0.0):
        """Returns dict(out_points, label, index, cls, cell).  Mutates the map layers."""
        assert cloud.dtype == POINT_DTYPE
        cloud = np.ascontiguousarray(cloud)
        n = cloud.shape[0]
        out = np.zeros(max(n, 1), dtype=np.uint8).repeat(32).reshape(-1, 32).view(POINT_DTYPE).reshape(-1)
        label = np.zeros(max(n, 1), dtype=np.uint8)
        index = np.zeros(max(n, 1), dtype=np.int32)
        cls = np.zeros(max(n, 1), dtype=np.uint8)
        cell = np.zeros(max(n, 1), dtype=np.int32)
        org = (C.c_float * 3)(*origin)
        out_n = self._L.ggo_filter_cloud(
            self._m, C.byref(self.cfg), cloud.ctypes.data, n, org, float(base_z),
            out.ctypes.data, label.ctypes.data, index.ctypes.data, cls.ctypes.data, cell.ctypes.data,
        )
        # NB: no .copy() -- numpy copies padded structured dtypes field by field and leaves the padding undefined
        return dict(out_points=out[:out_n], label=label[:n], index=index[:n], cls=cls[:n], cell=cell[:n])

    # stage-wise entry points (for per-kernel parity tests)
    def stage_reset(self):
        self._L.ggo_stage_reset(self._m)

    def stage_insert(self, cloud: np.ndarray, origin=(0.0, 0.0, 0.0)):
        cloud = np.ascontiguousarray(cloud)
        n = cloud.shape[0]
        cls = np.zeros(max(n, 1), dtype=np.uint8)
        cell = np.zeros(max(n, 1), dtype=np.int32)
        org = (C.c_float * 3)(*origin)
        self._L.ggo_stage_insert(self._m, C.byref(self.cfg), cloud.ctypes.data, n, org, cls.ctypes.data, cell.ctypes.data)
        return cls[:n], cell[:n]

    def stage_detect(self):
        self._L.ggo_stage_detect(self._m, C.byref(self.cfg))

    def stage_spiral(self, base_z=0.0):
        self._L.ggo_stage_spiral(self._m, C.byref(self.cfg), float(base_z))

    # the public stage members of include/groundgrid/GroundSegmentation.h:59-62, one by one
    def stage_detect_section(self, section: int):
        """detect_ground_patches(map, section): variance := m2 ./ (points + FLT_MIN), then one quadrant (0..3)."""
        self._L.ggo_stage_detect_section(self._m, C.byref(self.cfg), int(section))

    def detect_ground_patch(self, S: int, i: int, j: int):
        self._L.ggo_detect_ground_patch(self._m, C.byref(self.cfg), int(S), int(i), int(j))

    def interpolate_cell(self, x: int, y: int):
        self._L.ggo_interpolate_cell(self._m, C.byref(self.cfg), int(x), int(y))


def tree_sum(vals) -> float:
    a = np.ascontiguousarray(vals, dtype=np.float32)
    return float(lib().ggo_tree_sum(a.ctypes.data_as(C.POINTER(C.c_float)), a.size))


def hypotf(x, y) -> float:
    return float(lib().ggo_hypotf(C.c_float(x), C.c_float(y)))


def spiral_visit_count(rows: int) -> int:
    return int(lib().ggo_spiral_visit_count(rows))
