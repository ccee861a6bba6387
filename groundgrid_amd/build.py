"""In-tree build of libgroundgrid_hip.so (hipcc --offload-arch=gfx950).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libgroundgrid_hip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "groundgrid_hip.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, jobs: int = 8) -> str:
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    if force or needs_build():
        subprocess.check_call(["make", "-C", CSRC, f"-j{jobs}"])
    return LIB
