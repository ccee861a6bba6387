"""The addressing of the tile-blocked per-call layers (gg_internal.h percall_index / percall_position / live_bit), checked on the
host: the header's functions are __host__ __device__, so a host-only compile of a small program exercises the very code the kernels
use."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_per_call_layer_addressing_is_a_bijection():
    src = os.path.join(ROOT, "tests", "cpp", "test_layout.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "t")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-x", "hip", "--cuda-host-only", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "groundgrid_amd", "csrc"), src, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    assert out.strip() == "ok", out
