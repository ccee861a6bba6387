"""CPU checks of the C-ABI boundary: the library loads, exports every symbol include/groundgrid_hip.h
declares, agrees with the header on struct sizes, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from groundgrid_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "groundgrid_hip.h")


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gg_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_lists():
    assert declared_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (gg_[a-z0-9_]+)", out))
    missing = [s for s in declared_functions() if s not in exported]
    assert not missing, missing
    for s in _lib.SYMBOLS:
        assert hasattr(lib, s)


def test_abi_version_and_kernel_names(lib):
    # v2: gg_move_map takes matrix entries, conventions, async host call; v3: gg_batch.slots; v4: gg_comm_init_rank_for;
    # v5: gg_batch.d_out_pc2, gg_run_stage, gg_filter_cloud_pc2_out, gg_get_gridmap_message, gg_device_error
    assert lib.gg_abi_version() == 6
    names = [lib.gg_kernel_name(k).decode() for k in range(_lib.GG_NUM_KERNELS)]
    assert names == ["k_classify", "k_scan", "k_scatter", "k_reduce", "k_patch", "k_sweep", "k_label"]


def test_struct_layouts_match_the_header():
    assert C.sizeof(_lib.GGGeometry) == 16
    assert C.sizeof(_lib.GGConfig) == 104          # 2 int, 11 double, 1 int (+pad) as the C compiler lays it out
    assert C.sizeof(_lib.GGBatch) == 120           # ABI v3: + slots, v5: + d_out_pc2
    assert C.sizeof(_lib.GGStageArgs) == 24 and C.sizeof(_lib.GGGridMapHeader) == 32
    assert C.sizeof(_lib.GGConventions) == 32
    from groundgrid_amd import api, synth
    assert synth.POINT_DTYPE.itemsize == 32 and api.POINT16_DTYPE.itemsize == 16
    # compile a tiny C program against the header and compare sizeof / offsetof
    prog = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "groundgrid_hip.h"
    int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %d %zu %zu %zu %zu\n", sizeof(gg_point32), sizeof(gg_point16), sizeof(gg_config),
        sizeof(gg_geometry), sizeof(gg_batch), offsetof(gg_point32, ring), offsetof(gg_batch, d_out_counts),
        sizeof(gg_conventions), GG_ASYNC_DEPTH, offsetof(gg_batch, d_out_pc2), sizeof(gg_stage_args), offsetof(gg_stage_args, base_z),
        sizeof(gg_gridmap_header)); return 0; }
    '''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        vals = list(map(int, subprocess.check_output([os.path.join(d, "t")], text=True).split()))
    assert vals == [32, 16, C.sizeof(_lib.GGConfig), 16, C.sizeof(_lib.GGBatch), 20, _lib.GGBatch.d_out_counts.offset,
                    C.sizeof(_lib.GGConventions), _lib.GG_ASYNC_DEPTH, _lib.GGBatch.d_out_pc2.offset, C.sizeof(_lib.GGStageArgs),
                    _lib.GGStageArgs.base_z.offset, C.sizeof(_lib.GGGridMapHeader)]


def test_defaults_are_the_reference_cfg(lib):
    c = _lib.GGConfig()
    lib.gg_default_config(C.byref(c))
    # cfg/GroundGrid.cfg:8-21
    assert (c.point_count_cell_variance_threshold, c.max_ring, c.thread_count) == (10, 1024, 8)
    assert (c.distance_factor, c.minimum_distance_factor) == (0.0001, 0.0005)
    assert (c.miminum_point_height_threshold, c.minimum_point_height_obstacle_threshold) == (0.3, 0.1)
    assert (c.outlier_tolerance, c.ground_patch_detection_minimum_point_count_threshold) == (0.1, 0.25)
    assert (c.patch_size_change_distance, c.occupied_cells_decrease_factor) == (20.0, 5.0)
    assert (c.occupied_cells_point_count_factor, c.min_outlier_detection_ground_confidence) == (20.0, 1.25)
    g = _lib.GGGeometry()
    lib.gg_default_geometry(C.byref(g))
    assert (g.length, g.resolution, g.min_dist_squared) == (120.0, np.float32(0.33), 12.0)
    assert g.vertical_point_ang_dist == np.float32(0.00174532925 * 2)


def test_no_gpu_means_loud_failure_not_fallback(lib, gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    g = _lib.GGGeometry(120.0, 0.33, 0.0, 0.0)
    rc = lib.gg_create(C.byref(g), 1, 1000, 0, C.byref(ctx))
    assert rc == -6 and not ctx.value  # GG_ERR_NO_DEVICE
    from groundgrid_amd import api
    with pytest.raises(_lib.GroundGridError):
        api.GroundSegmentation().init(120.0, 0.33)


def test_bad_arguments_are_rejected_before_touching_the_device(lib):
    ctx = C.c_void_p()
    g = _lib.GGGeometry(120.0, 0.33, 0.0, 0.0)
    assert lib.gg_create(C.byref(g), 0, 1000, 0, C.byref(ctx)) == -1
    assert lib.gg_create(C.byref(g), 1, 0, 0, C.byref(ctx)) == -1
    assert lib.gg_create(C.byref(g), 1, 1000, 0, None) == -1
    bad = _lib.GGGeometry(100.4, 0.4, 0.0, 0.0)  # grid_map says 251 cells, init() says 250
    assert lib.gg_create(C.byref(bad), 1, 1000, 0, C.byref(ctx)) == -2
    assert lib.gg_set_config(None, None) == -1
    assert lib.gg_filter_batch(None, None, None) == -1
    assert lib.gg_get_layer(None, 0, 0, None) == -5
    assert lib.gg_set_conventions(None, None) == -1
    assert lib.gg_move_map(None, 0, 0.0, 0.0, None, None) == -5
    assert lib.gg_filter_cloud_wait(None, 0, None, None, None, None) == -1
    q = (C.c_double * 4)(0, 0, 0, 1)
    R = (C.c_double * 9)()
    assert lib.gg_rotation_from_quaternion(7, q, R) == -1 and lib.gg_rotation_from_quaternion(0, None, R) == -1


def test_collective_entry_points_reject_bad_arguments(lib):
    """gg_allgather_label_masks & co (the C-ABI face of BASELINE configs[2]'s all-gather): exported, and null communicators /
    contexts / buffers are refused before RCCL or the device is touched."""
    assert lib.gg_collective_available() in (0, 1)   # librccl.so ships with ROCm; it is bound with dlopen on first use
    assert lib.gg_allgather_label_masks(None, None, None, None, 16, None) == -1
    assert lib.gg_comm_unique_id(None) == -1
    comm = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert lib.gg_comm_init_rank(ident, 0, 0, C.byref(comm)) == -1 and lib.gg_comm_init_rank(ident, 2, 2, C.byref(comm)) == -1
    assert lib.gg_comm_init_rank(None, 1, 0, C.byref(comm)) == -1 and lib.gg_comm_destroy(None) == -1


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under groundgrid_amd/ or include/ may reference it."""
    for base in ("groundgrid_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "import oracle" not in txt and "from oracle" not in txt and "gg_oracle.h\"" not in txt.replace("oracle/gg_oracle.c", ""), f
