"""Compile check of the reference-typed binding (groundgrid_amd/host/ros/GroundSegmentationHip.cpp): our definitions of
groundgrid::GroundSegmentation's members against the reference's OWN class declaration (read from /root/reference, never
copied) plus declaration-only stand-ins of the ROS / PCL / grid_map types (tests/cpp/decl_only/README.md).  Compile only --
nothing of the reference is built, nothing is linked or run, no parity is pinned by this."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INCLUDE = "/root/reference/include"
SRC = os.path.join(ROOT, "groundgrid_amd", "host", "ros", "GroundSegmentationHip.cpp")


@pytest.mark.skipif(not os.path.isdir(REF_INCLUDE), reason="the reference checkout only exists in the build container")
def test_binding_compiles_against_the_reference_class_declaration(tmp_path):
    obj = str(tmp_path / "binding.o")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", SRC, "-o", obj,
                           "-I", os.path.join(ROOT, "tests", "cpp", "decl_only"), "-I", REF_INCLUDE, "-I", os.path.join(ROOT, "include")])
    syms = subprocess.check_output(["nm", "-C", "--defined-only", obj], text=True)
    # every member function the reference header declares (include/groundgrid/GroundSegmentation.h:53-62) is defined here
    for name in ("init(ros::NodeHandle&, unsigned long, float const&)", "setConfig(groundgrid::GroundGridConfig const&)",
                 "filter_cloud(", "insert_cloud(", "detect_ground_patches(grid_map::GridMap&, unsigned short) const",
                 "void groundgrid::GroundSegmentation::detect_ground_patch<3>(", "void groundgrid::GroundSegmentation::detect_ground_patch<5>(",
                 "spiral_ground_interpolation(", "interpolate_cell("):
        assert re.search(r"groundgrid::GroundSegmentation::" + re.escape(name.split("groundgrid::GroundSegmentation::")[-1]), syms), name
    undefined = subprocess.check_output(["nm", "-C", "--undefined-only", obj], text=True)
    for entry in ("gg_create", "gg_set_config", "gg_filter_cloud_layers", "gg_run_stage", "gg_get_layers", "gg_set_layer", "gg_set_map_position", "gg_insert_cloud", "gg_abi_version"):
        assert re.search(r"\bU " + entry + r"\b", undefined), entry  # ... and forwards to the C ABI


@pytest.mark.skipif(not os.path.isdir(REF_INCLUDE), reason="the reference checkout only exists in the build container")
def test_map_manager_binding_compiles_against_the_reference_class_declaration(tmp_path):
    """groundgrid_amd/host/ros/GroundGridHip.cpp -- GroundGrid::initGroundGrid / update on the device (gg_reset_map / gg_move_map)
    -- against the reference's own include/groundgrid/GroundGrid.h:50-85."""
    obj = str(tmp_path / "groundgrid_binding.o")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", os.path.join(ROOT, "groundgrid_amd", "host", "ros", "GroundGridHip.cpp"), "-o", obj,
                           "-I", os.path.join(ROOT, "tests", "cpp", "decl_only"), "-I", REF_INCLUDE, "-I", os.path.join(ROOT, "include")])
    syms = subprocess.check_output(["nm", "-C", "--defined-only", obj], text=True)
    for name in ("GroundGrid::GroundGrid()", "GroundGrid::~GroundGrid()", "GroundGrid::setConfig(groundgrid::GroundGridConfig&)",
                 "GroundGrid::initGroundGrid(", "GroundGrid::update("):
        assert "groundgrid::" + name in syms, name
    undefined = subprocess.check_output(["nm", "-C", "--undefined-only", obj], text=True)
    for entry in ("gg_reset_map", "gg_move_map", "gg_rotation_from_quaternion", "gg_get_map_position"):
        assert re.search(r"\bU " + entry + r"\b", undefined), entry


def test_stand_ins_define_no_functions():
    """Declaration-only means it: no function bodies in the stand-in headers (data members and macros only)."""
    base = os.path.join(ROOT, "tests", "cpp", "decl_only")
    for dp, _, files in os.walk(base):
        for f in files:
            if f.endswith((".h", ".hpp")):
                txt = re.sub(r"//.*", "", open(os.path.join(dp, f)).read())
                assert not re.search(r"\)\s*(const)?\s*\{", txt), os.path.join(dp, f)


def test_integration_doc_matches_the_code():
    import glob
    import json
    import re

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "GG_SPIRAL_CAPS" not in doc and "GG_DEBUG_SCHEDULE" not in doc  # knobs of the retired level-scheduled sweep
    assert "GroundSegmentationHip.cpp" in doc and "base_plane" in doc  # ABI v2 move_map in the device-resident binding
    # the default layer masks of the two bindings, as the code has them
    core = open(os.path.join(ROOT, "groundgrid_amd", "host", "binding_core.hpp")).read()
    shell = open(os.path.join(ROOT, "groundgrid_amd", "host", "ros", "GroundSegmentationHip.cpp")).read()
    assert re.search(r"layers_from_env\(core->device_resident\(\) \? groundgrid_hip::LAYERS_STATE : groundgrid_hip::LAYERS_ALL\)", shell)
    assert "LAYERS_STATE = (1u << GG_LAYER_GROUND) | (1u << GG_LAYER_GROUNDPATCH) | (1u << GG_LAYER_POINTS) | (1u << GG_LAYER_POINTSRAW)" in core
    assert "Default `state` for a device-resident map" in doc and "`all` for a host-managed one" in doc
    assert "`state` (ground, groundpatch, points, pointsRaw)" in doc
    assert "Default `none`" not in doc
    # entry points and flags of ABI v6 are named
    header = open(os.path.join(ROOT, "include", "groundgrid_hip.h")).read()
    for name in ("gg_insert_cloud", "GG_FLAG_EAGER_LAYERS", "GG_FLAG_CONCURRENT_HALVES", "GG_FLAG_MINIMAL_LAYERS", "gg_batch_fence"):
        assert name in header and name in doc, name
    # the quoted host_api figures are those of the last committed profile round
    rounds = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default.json")))
    assert rounds
    last = rounds[-1]
    line = json.loads(open(last).read().strip().splitlines()[-1])
    api = line["host_api"]
    assert os.path.basename(os.path.dirname(last)) in doc, "INTEGRATION.md names another profile round than the last committed one"
    quoted = {"binding_like_ms": api["binding_like_ms"], "binding_like_two_calls_ms": api["binding_like_two_calls_ms"], "sync_ms": api["sync_ms"]}
    for key, ms in quoted.items():
        assert f"{ms:.2f}" in doc, (key, ms)
    assert f"{api['binding_like_clouds_per_s']:.0f}" in doc and f"{api['device_resident_binding_clouds_per_s']:.0f}" in doc