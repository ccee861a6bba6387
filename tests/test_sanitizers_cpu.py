"""Sanitizer runs of the CPU-side code (the reference was never run under a sanitizer; its insertion threads race,
src/GroundSegmentation.cpp:101-106).

* the oracle (oracle/gg_oracle.c, the restatement every parity test hangs on) built with AddressSanitizer + UndefinedBehaviourSanitizer
  and driven over the committed golden vectors, the edge-case clouds (NaN / inf / signalling NaN / far-away points, empty cloud,
  everything outside), a line-of-sight-heavy cloud and a moving map -- any out-of-bounds access or undefined operation aborts;
* the host-side helper threads of the library (groundgrid_amd/csrc/host_helper.h: packing the input cloud and assembling the
  returned cloud in parallel parts) built WITHOUT HIP under ThreadSanitizer, and under ASan + UBSan with the fork() case.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GCC = shutil.which("gcc")
GXX = shutil.which("g++")


def _runtime(name):
    p = subprocess.run([GCC, f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


ORACLE_DRIVER = r'''
import os, sys
import numpy as np
sys.path.insert(0, ROOT)
from groundgrid_amd import synth
from oracle import oracle
golden = os.path.join(ROOT, "tests", "golden")
checked = 0
for fname in sorted(f for f in os.listdir(golden) if f.endswith(".npz")):
    g = np.load(os.path.join(golden, fname))
    cloud = np.frombuffer(g["cloud"].tobytes(), dtype=synth.POINT_DTYPE)
    m = oracle.OracleMap(float(g["length"]), float(g["resolution"]), pos=tuple(g["pos"]))
    for f in range(int(g["frames"])):
        r = m.filter_cloud(cloud, tuple(g["origin"]), float(g["base_z"]))
        assert np.array_equal(r["label"], g[f"label_{f}"]) and np.array_equal(r["index"], g[f"index_{f}"]) and np.array_equal(r["cls"], g[f"cls_{f}"]), fname
        for layer in ("ground", "groundpatch", "variance"):
            assert np.array_equal(m.layer(layer), g[f"{layer}_{f}"], equal_nan=True), (fname, layer)
        checked += 1
# edge cases (tests/test_gpu_parity.py::test_edge_cases) incl. a signalling NaN height
pts = np.array([[5, 5, -1], [1, 1, -1], [5, 5, -1], [500, 0, -1], [np.nan, 0, -1], [5, 5, np.nan], [-59.9, -59.9, -1], [np.inf, 1, 0],
                [59.99, 59.99, 0.5], [0, 0, 3], [3, -59.5, -1.6], [5, 5, -np.inf], [-1e30, 2, 0], [5, 5.01, 0.0]], dtype=np.float32)
cloud = synth.make_cloud(pts, ring=[0, 0, 2000, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6])
cloud["z"].view(np.uint32)[13] = 0x7FA00000
for c in (cloud, synth.empty_cloud(0), synth.make_cloud(np.array([[900.0, 900.0, 0.0]], dtype=np.float32))):
    m = oracle.OracleMap(120.0, 0.33)
    for _ in range(2):
        m.filter_cloud(c, (0.0, 0.0, 0.0), -1.73)
# a corrupt height far below the map (bounded walk), many line-of-sight candidates, a shifted origin
base = synth.hdl64_cloud(seed=12, n_az=300)
low = synth.clone_cloud(base)
rng = np.random.default_rng(12)
sel = rng.random(len(low)) < 0.2
low["z"][sel] -= rng.uniform(0.3, 2.5, sel.sum()).astype(np.float32)
low["z"][7] = -1e9
m = oracle.OracleMap(120.0, 0.33)
n_out = 0
for c, origin in ((base, (0.0, 0.0, 0.0)), (low, (0.0, 0.0, 0.0)), (low, (7.5, -3.0, 0.4))):
    n_out += int((m.filter_cloud(c, origin, -1.73)["cls"] == oracle.OUTLIER).sum())
assert n_out > 100
# odd grid sizes, a tiny map, the dense geometry, a moving map
for length, res in ((4.0, 0.33), (33.0, 0.33), (61.0, 0.25)):
    c = synth.clone_cloud(base)
    c["x"] *= np.float32(length / 120.0); c["y"] *= np.float32(length / 120.0)
    m = oracle.OracleMap(length, res)
    m.filter_cloud(c, (0.0, 0.0, 0.0), -1.73); m.filter_cloud(c, (0.0, 0.0, 0.0), -1.73)
m = oracle.OracleMap(120.0, 0.33)
for k in range(5):  # the map follows the vehicle (GroundGrid::update): shifts in both directions, exposed rows / columns re-seeded
    x, y = 1.3 * k * (1 if k < 3 else -1), -0.9 * k
    m.update(x, y, (-x, -y, 1.73, 0.0, 0.0, np.sin(0.05 * k), np.cos(0.05 * k)))
    m.filter_cloud(base, (x, y, 0.0), -1.73)
print("oracle under sanitizers OK:", checked, "golden frames")
'''


@pytest.mark.skipif(not GCC, reason="no gcc")
def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    asan = _runtime("libasan.so")
    if not asan:
        pytest.skip("this gcc has no libasan")
    lib = str(tmp_path / "libgg_oracle_san.so")
    subprocess.check_call([GCC, "-O1", "-g", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fexcess-precision=standard",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-shared", "-o", lib,
                           os.path.join(ROOT, "oracle", "gg_oracle.c"), "-lm", "-lpthread"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               GG_ORACLE_LIB=lib)
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + ORACLE_DRIVER], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert "oracle under sanitizers OK" in p.stdout


@pytest.mark.skipif(not GXX, reason="no g++")
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_helper_threads_under_sanitizers(sanitizer, tmp_path):
    """host_helper.h alone (no HIP): many split / pack / assemble rounds with 0, 1, 3 and 7 helpers, two helper sets side by side
    and -- outside ThreadSanitizer, which cannot follow a multi-threaded fork -- a fork()ed child that must start its own helpers."""
    exe = str(tmp_path / "test_host_helper")
    subprocess.check_call([GXX, "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-ffp-contract=off",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "groundgrid_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_host_helper.cpp"), "-o", exe, "-lpthread"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "ThreadSanitizer" not in p.stderr and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
    assert "host helper OK" in p.stdout
