"""ctypes loader for libgroundgrid_hip.so (the C ABI of include/groundgrid_hip.h).

The library is built in-tree by ``groundgrid_amd.build.build()`` (hipcc, gfx950).  Loading fails loudly
if the .so is missing: the product has no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GROUNDGRID_HIP_LIB") or os.path.join(_HERE, "libgroundgrid_hip.so")  # override: side-by-side builds

GG_OK = 0
STATUS = {
    0: "GG_OK",
    -1: "GG_ERR_INVALID",
    -2: "GG_ERR_GEOMETRY",
    -3: "GG_ERR_NOMEM",
    -4: "GG_ERR_HIP",
    -5: "GG_ERR_CAPACITY",
    -6: "GG_ERR_NO_DEVICE",
}

GG_ABI_VERSION = 6  # include/groundgrid_hip.h
GG_POINT32, GG_POINT16 = 0, 1
GG_FLAG_MINIMAL_LAYERS, GG_FLAG_PROFILE, GG_FLAG_CONCURRENT_HALVES, GG_FLAG_EAGER_LAYERS = 1, 2, 4, 8
GG_NUM_KERNELS = 7
GG_NUM_LAYERS = 11

LAYERS = [
    "points", "ground", "groundpatch", "minGroundHeight", "maxGroundHeight", "groundCandidates",
    "planeDist", "m2", "meanVariance", "pointsRaw", "variance",
]

# every symbol include/groundgrid_hip.h declares
SYMBOLS = [
    "gg_abi_version", "gg_kernel_name", "gg_default_config", "gg_default_geometry", "gg_create", "gg_destroy",
    "gg_set_config", "gg_get_config", "gg_set_flags", "gg_get_size", "gg_get_geometry", "gg_last_error",
    "gg_reset_map", "gg_reset_maps", "gg_set_map_position", "gg_move_map", "gg_get_map_position", "gg_set_layer", "gg_get_layer", "gg_get_layers", "gg_get_expected_points",
    "gg_filter_cloud", "gg_filter_cloud_tf", "gg_filter_cloud_pc2", "gg_get_layer_image_u8", "gg_get_terrain_image", "gg_filter_batch", "gg_synchronize", "gg_get_point_classes", "gg_get_kernel_times",
    "gg_set_conventions", "gg_get_conventions", "gg_rotation_from_quaternion", "gg_transform_from_pose",
    "gg_filter_cloud_async", "gg_filter_cloud_wait", "gg_debug_emulate_ring_sweep", "gg_debug_sweep_sync_selftest",
    "gg_batch_fence", "gg_device_error", "gg_filter_cloud_layers", "gg_host_register", "gg_host_unregister", "gg_run_stage", "gg_insert_cloud", "gg_filter_cloud_pc2_out", "gg_get_gridmap_message",
    "gg_collective_available", "gg_comm_unique_id", "gg_comm_init_rank", "gg_comm_init_rank_for", "gg_comm_destroy", "gg_allgather_label_masks",
]

GG_EIGEN_33, GG_EIGEN_34_SSE = 0, 1
GG_ROT_TF2, GG_ROT_KDL = 0, 1
ROTATION = {"tf2": GG_ROT_TF2, "kdl": GG_ROT_KDL}
GG_ASYNC_DEPTH = 2
GG_STREAM_DEFAULT = C.c_void_p(-1).value  # include/groundgrid_hip.h: the legacy default ("null") stream -- what torch's handle 0 means


class GGConfig(C.Structure):
    """gg_config == groundgrid::GroundGridConfig (cfg/GroundGrid.cfg:8-21)"""

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


class GGConventions(C.Structure):
    _fields_ = [("eigen_reduction", C.c_int), ("reserved", C.c_int * 7)]


class GGGeometry(C.Structure):
    _fields_ = [
        ("length", C.c_float),
        ("resolution", C.c_float),
        ("vertical_point_ang_dist", C.c_float),
        ("min_dist_squared", C.c_float),
    ]


class GGBatch(C.Structure):
    _fields_ = [
        ("n_clouds", C.c_int),
        ("first_slot", C.c_int),
        ("point_format", C.c_int),
        ("d_points", C.c_void_p),
        ("cloud_stride", C.c_size_t),
        ("n_points", C.POINTER(C.c_int32)),
        ("origins", C.POINTER(C.c_float)),
        ("base_z", C.POINTER(C.c_double)),
        ("transforms", C.POINTER(C.c_double)),
        ("d_labels", C.c_void_p),
        ("d_out_index", C.c_void_p),
        ("d_out_clouds", C.c_void_p),
        ("d_out_counts", C.c_void_p),
        ("d_label_masks", C.c_void_p),
        ("slots", C.POINTER(C.c_int32)),
        ("d_out_pc2", C.c_void_p),
    ]


GG_PC2_POINT_STEP = 18
GG_STAGE_DETECT_GROUND_PATCHES, GG_STAGE_SPIRAL_GROUND_INTERPOLATION = 1, 2
GG_STAGE_DETECT_GROUND_PATCH_3, GG_STAGE_DETECT_GROUND_PATCH_5, GG_STAGE_INTERPOLATE_CELL = 3, 4, 5


class GGStageArgs(C.Structure):
    _fields_ = [("section", C.c_int), ("i", C.c_int), ("j", C.c_int), ("base_z", C.c_double)]


class GGGridMapHeader(C.Structure):
    _fields_ = [("seq", C.c_uint32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32), ("frame_id", C.c_char_p), ("basic_layers", C.c_uint)]


class GroundGridError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GroundGridError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    L.gg_abi_version.restype = C.c_int
    if L.gg_abi_version() != GG_ABI_VERSION:  # (struct layouts below are this version's: a stale .so would read past gg_batch)
        raise GroundGridError(f"{LIB_PATH} has ABI version {L.gg_abi_version()}, this package expects {GG_ABI_VERSION}: rebuild it")
    L.gg_kernel_name.restype = C.c_char_p
    L.gg_kernel_name.argtypes = [C.c_int]
    L.gg_default_config.argtypes = [P(GGConfig)]
    L.gg_default_config.restype = None
    L.gg_default_geometry.argtypes = [P(GGGeometry)]
    L.gg_default_geometry.restype = None
    L.gg_create.argtypes = [P(GGGeometry), C.c_int, C.c_size_t, C.c_int, P(vp)]
    L.gg_destroy.argtypes = [vp]
    L.gg_destroy.restype = None
    L.gg_set_config.argtypes = [vp, P(GGConfig)]
    L.gg_get_config.argtypes = [vp, P(GGConfig)]
    L.gg_set_flags.argtypes = [vp, C.c_uint]
    L.gg_get_size.argtypes = [vp, P(C.c_int), P(C.c_int)]
    L.gg_get_geometry.argtypes = [vp, P(C.c_double), P(C.c_double), P(C.c_double)]
    L.gg_last_error.argtypes = [vp]
    L.gg_last_error.restype = C.c_char_p
    L.gg_reset_map.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_float]
    L.gg_reset_maps.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float, C.c_int, vp]
    L.gg_set_map_position.argtypes = [vp, C.c_int, C.c_double, C.c_double]
    L.gg_move_map.argtypes = [vp, C.c_int, C.c_double, C.c_double, P(C.c_double), P(C.c_int)]
    L.gg_get_map_position.argtypes = [vp, C.c_int, P(C.c_double), P(C.c_double)]
    L.gg_set_layer.argtypes = [vp, C.c_int, C.c_int, vp]
    L.gg_get_layer.argtypes = [vp, C.c_int, C.c_int, vp]
    L.gg_get_expected_points.argtypes = [vp, vp]
    L.gg_filter_cloud.argtypes = [vp, C.c_int, vp, C.c_size_t, P(C.c_float), C.c_double, vp, P(C.c_size_t), vp, vp]
    L.gg_filter_cloud_tf.argtypes = [vp, C.c_int, vp, C.c_size_t, P(C.c_double), P(C.c_float), C.c_double, vp, P(C.c_size_t), vp, vp]
    L.gg_filter_cloud_pc2.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, P(C.c_double), P(C.c_float), C.c_double, vp, vp, P(C.c_size_t)]
    L.gg_get_layers.argtypes = [vp, C.c_int, P(vp)]
    L.gg_filter_cloud_pc2_out.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, P(C.c_double), P(C.c_float), C.c_double, vp, P(C.c_size_t)]
    L.gg_get_gridmap_message.argtypes = [vp, C.c_int, C.c_uint, P(GGGridMapHeader), vp, C.c_size_t, P(C.c_size_t)]
    L.gg_run_stage.argtypes = [vp, C.c_int, C.c_int, P(GGStageArgs)]
    L.gg_insert_cloud.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, P(C.c_float), vp, vp]
    L.gg_filter_cloud_layers.argtypes = [vp, C.c_int, vp, C.c_size_t, P(C.c_double), P(C.c_float), C.c_double, vp, P(C.c_size_t), vp, vp, P(vp)]
    L.gg_host_register.argtypes = [vp, vp, C.c_size_t]
    L.gg_host_unregister.argtypes = [vp, vp]
    L.gg_get_layer_image_u8.argtypes = [vp, C.c_int, C.c_int, vp, P(C.c_float), P(C.c_float)]
    L.gg_get_terrain_image.argtypes = [vp, C.c_int, vp]
    L.gg_filter_batch.argtypes = [vp, P(GGBatch), vp]
    L.gg_synchronize.argtypes = [vp]
    L.gg_device_error.argtypes = [vp, C.c_int]
    L.gg_batch_fence.argtypes = [vp, vp]
    L.gg_get_point_classes.argtypes = [vp, C.c_int, C.c_size_t, vp, vp]
    L.gg_get_kernel_times.argtypes = [vp, P(C.c_double), P(C.c_int64), C.c_int]
    L.gg_set_conventions.argtypes = [vp, P(GGConventions)]
    L.gg_get_conventions.argtypes = [vp, P(GGConventions)]
    L.gg_rotation_from_quaternion.argtypes = [C.c_int, P(C.c_double), P(C.c_double)]
    L.gg_transform_from_pose.argtypes = [C.c_int, P(C.c_double), P(C.c_double)]
    L.gg_filter_cloud_async.argtypes = [vp, C.c_int, vp, C.c_size_t, P(C.c_double), P(C.c_float), C.c_double, P(C.c_int)]
    L.gg_filter_cloud_wait.argtypes = [vp, C.c_int, vp, P(C.c_size_t), vp, vp]
    L.gg_collective_available.argtypes = []
    L.gg_comm_unique_id.argtypes = [vp]
    L.gg_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, P(vp)]
    L.gg_comm_init_rank_for.argtypes = [vp, vp, C.c_int, C.c_int, P(vp)]
    L.gg_comm_destroy.argtypes = [vp]
    L.gg_allgather_label_masks.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    _lib = L
    return L
