"""The C++ host adapter (groundgrid_amd/host/GroundSegmentation.hpp): compiles and links on CPU; on the GPU box the
compiled program runs the reference-style call sequence and compares with the C oracle bit for bit."""
import os
import subprocess

import pytest

from groundgrid_amd import build
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_adapter")


def compile_adapter():
    build.build()
    oracle.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        subprocess.check_call([
            "g++", "-O1", "-std=c++17", src, "-o", EXE,
            "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "groundgrid_amd", "host"), "-I", os.path.join(ROOT, "oracle"),
            "-L", os.path.join(ROOT, "groundgrid_amd"), "-lgroundgrid_hip", "-L", os.path.join(ROOT, "oracle"), "-lgg_oracle",
            "-Wl,-rpath," + os.path.join(ROOT, "groundgrid_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
            "-Wl,-rpath,/opt/rocm/lib",
        ])
    return EXE


def test_adapter_compiles_and_links_against_the_c_abi():
    assert os.path.exists(compile_adapter())


@pytest.mark.gpu
def test_adapter_matches_oracle_on_gpu():
    exe = compile_adapter()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
