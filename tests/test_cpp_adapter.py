"""The C++ host adapter (groundgrid_amd/host/GroundSegmentation.hpp): compiles and links on CPU; on the GPU box the
compiled program runs the reference-style call sequence and compares with the C oracle bit for bit."""
import os
import subprocess

import pytest

from groundgrid_amd import build
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_adapter")
# every program below is compiled against these: a struct that grows in the C ABI (gg_batch in ABI v5) must rebuild them all
HEADERS = [os.path.join(ROOT, "include", "groundgrid_hip.h"), os.path.join(ROOT, "groundgrid_amd", "host", "GroundSegmentation.hpp"),
           os.path.join(ROOT, "groundgrid_amd", "host", "binding_core.hpp"), os.path.join(ROOT, "oracle", "gg_oracle.h")]


def stale(exe, *srcs):
    return not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in list(srcs) + HEADERS)


def compile_adapter():
    build.build()
    oracle.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp")
    if stale(EXE, src):
        subprocess.check_call([
            "g++", "-O1", "-std=c++17", src, "-o", EXE,
            "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "groundgrid_amd", "host"), "-I", os.path.join(ROOT, "oracle"),
            "-L", os.path.join(ROOT, "groundgrid_amd"), "-lgroundgrid_hip", "-L", os.path.join(ROOT, "oracle"), "-lgg_oracle",
            "-Wl,-rpath," + os.path.join(ROOT, "groundgrid_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
            "-Wl,-rpath,/opt/rocm/lib",
        ])
    return EXE


BINDING_EXE = os.path.join(ROOT, "tests", "cpp", "test_binding_core")


def compile_binding_core():
    build.build()
    oracle.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_binding_core.cpp")
    hdr = os.path.join(ROOT, "groundgrid_amd", "host", "binding_core.hpp")
    if stale(BINDING_EXE, src, hdr):
        subprocess.check_call([
            "g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", src, "-o", BINDING_EXE,
            "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "groundgrid_amd", "host"), "-I", os.path.join(ROOT, "oracle"),
            "-L", os.path.join(ROOT, "groundgrid_amd"), "-lgroundgrid_hip", "-L", os.path.join(ROOT, "oracle"), "-lgg_oracle",
            "-Wl,-rpath," + os.path.join(ROOT, "groundgrid_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
            "-Wl,-rpath,/opt/rocm/lib",
        ])
    return BINDING_EXE


def test_binding_core_compiles_and_links_against_the_c_abi():
    assert os.path.exists(compile_binding_core())


@pytest.mark.gpu
def test_two_objects_keep_separate_maps_host_managed_and_device_resident():
    """groundgrid_amd/host/binding_core.hpp -- the logic of both reference-typed bindings below the ROS types: one context per
    GroundSegmentation object, a host-managed map (uploads on moves, all layers down per cloud) next to a device-resident one
    (gg_reset_map / gg_move_map, nothing down during the drive, the context re-created for a larger cloud in mid-drive),
    interleaved, every returned cloud and every layer against the C oracle."""
    exe = compile_binding_core()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "binding core OK" in p.stdout


STREAMS_EXE = os.path.join(ROOT, "tests", "cpp", "test_streams")


def compile_streams():
    build.build()
    oracle.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_streams.c")
    if stale(STREAMS_EXE, src):
        subprocess.check_call([
            "gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", src, "-o", STREAMS_EXE,
            "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), "-I", "/opt/rocm/include",
            "-L", os.path.join(ROOT, "groundgrid_amd"), "-lgroundgrid_hip", "-L", os.path.join(ROOT, "oracle"), "-lgg_oracle",
            "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
            "-Wl,-rpath," + os.path.join(ROOT, "groundgrid_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
            "-Wl,-rpath,/opt/rocm/lib",
        ])
    return STREAMS_EXE


def test_plain_c_stream_and_async_program_compiles():
    assert os.path.exists(compile_streams())


@pytest.mark.gpu
def test_foreign_stream_ordering_and_async_pipeline_in_plain_c():
    """gg_filter_batch on a caller stream followed at once by gg_get_layer / gg_set_layer / gg_move_map, and the two-deep
    gg_filter_cloud_async pipeline: a C program with no Python in the loop, bit-compared with the C oracle."""
    exe = compile_streams()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr


def test_adapter_compiles_and_links_against_the_c_abi():
    assert os.path.exists(compile_adapter())


@pytest.mark.gpu
def test_adapter_matches_oracle_on_gpu():
    exe = compile_adapter()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
