#!/usr/bin/env python3
"""compare.py -- decide the oracle's third-party conventions from vectors printed by tools/pin/probe.cpp on a ROS box.

  python3 tools/pin/compare.py vectors.json        # report per section which oracle variant reproduces the vectors
  python3 tools/pin/compare.py --emit out.json [--eigen 0|1] [--rotation kdl|tf2]
                                                   # self-test: write the vectors the ORACLE computes under a variant

The inputs are regenerated here with a Python mirror of pin_inputs.h (checked against the C++ generator by
tests/test_pin_kit_cpu.py); every section is then evaluated with the oracle (oracle/gg_oracle.c through oracle/oracle.py)
under each variant it offers.  Exit status 0 = every section is reproduced bit for bit by some variant (the oracle is
pinned; the report names the variants to select with gg_set_conventions / rotation=), 1 = some section matches no variant
(a restated convention is wrong: fix the oracle and the device function it names).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402  (test infrastructure: this tool is a checker, not product code)

F32 = np.float32


class Lcg:
    """Mirror of tools/pin/pin_inputs.h."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFF

    def next(self) -> int:
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return self.s

    def wide_float(self) -> np.float32:
        m = (self.next() >> 8) - (1 << 23)
        e = (self.next() >> 27) - 12
        return F32(math.ldexp(float(m), min(e, 12) - 23))

    def unit(self) -> float:
        return (self.next() >> 8) / 16777216.0


def hex32(f) -> str:
    return "%08x" % struct.unpack("<I", struct.pack("<f", float(f)))[0]


def hex64(d) -> str:
    return "%016x" % struct.unpack("<Q", struct.pack("<d", float(d)))[0]


def block_values(M, S):
    n = M.shape[0]
    for j in range(0, n - S + 1, 3):
        for i in range(0, n - S + 1, 3):
            yield M[i:i + S, j:j + S].reshape(-1, order="F")  # column-major linear index, as Eigen's Block


def block_sum(e, S, order) -> float:
    oracle.set_eigen_reduction(order)
    v = np.ascontiguousarray(e, dtype=np.float32)
    return oracle.lib().ggo_block_sum(v.ctypes.data_as(C.POINTER(C.c_float)), S)


def eigen_section(order: int) -> dict:
    n = 41
    A, B = np.zeros((n, n), dtype=np.float32), np.zeros((n, n), dtype=np.float32)
    g = Lcg(0xE16E0001)
    for j in range(n):
        for i in range(n):
            A[i, j] = g.wide_float()
            B[i, j] = g.wide_float()
    out = {}
    for S in (3, 5):
        out[f"eigen_sum{S}"] = [hex32(block_sum(e, S, order)) for e in block_values(A, S)]
        prods = [hex32(block_sum((a * b).astype(np.float32), S, order)) for a, b in zip(block_values(A, S), block_values(B, S))]
        out[f"eigen_prod{S}"] = prods
        out[f"eigen_arrprod{S}"] = list(prods)  # .array().cwiseProduct().sum(): same traversal as the matrix form
        out[f"eigen_min{S}"] = [hex32(np.min(e)) for e in block_values(A, S)]
    P = np.abs(A)
    V = (B / (P + np.finfo(np.float32).tiny)).astype(np.float32)
    out["eigen_variance"] = [hex32(V[k % n, k // n]) for k in range(64)]
    oracle.set_eigen_reduction(0)
    return out


def c_rem(a: int, b: int) -> int:
    return int(math.fmod(a, b))  # C++ '%' truncates toward zero


def grid_section() -> dict:
    px, py = 12.34, -7.77
    m = oracle.OracleMap(120.0, 0.33, pos=(px, py))
    out = {"gm_size": [m.rows, m.cols],
           "gm_geometry": [hex64(m.resolution), hex64(m.length[0]), hex64(m.length[1]), hex64(m.position[0]), hex64(m.position[1])]}
    res, half = m.resolution, 0.5 * m.length[0]
    g = Lcg(0x61D00002)
    vals = []
    for k in range(-2, m.rows + 3):
        for v in range(3):
            x = (px + half) - float(k) * res
            if v == 1:
                x = math.nextafter(x, 1e300)
            if v == 2:
                x = math.nextafter(x, -1e300)
            y = (py + half) - (float(c_rem(k * 7, 364)) + g.unit()) * res
            inside, r, c = m.get_index(x, y)
            vals += [int(inside), int(inside), r, c]
            inside, r, c = m.get_index(y - py + px, x - px + py)
            vals += [int(inside), int(inside), r, c]
    out["gm_index"] = vals
    vals = []
    for _ in range(4000):
        rx = px + (g.unit() - 0.5) * 130.0
        ry = py + (g.unit() - 0.5) * 130.0
        inside, r, c = m.get_index(rx, ry)
        vals += [int(inside), r, c]
    out["gm_index_random"] = vals
    return out


MOVES = [(0.16, -0.16), (0.17, 0.0), (0.7, -0.34), (-3.0, 5.2), (-3.0, 5.2), (2.475, 5.2), (40.0, 5.2), (40.0, -90.0)]
POSES = [(0, 0, 1, 0, 0, 0, 1),
         (0.3, 0.2, 1.5, 0.02, -0.01, 0.3, 0.9533),
         (0.0, 0.0, 1.25, 0, 0, 0, 1),
         (1, 2, 3, 0, 0, 0.70710678118654757, 0.70710678118654757),
         (1, 2, 3, 0, 0, 0.70710678118654757, 0.70710678118654757),
         (-4.0, 1.0, 0.5, 0.0499791692706783, 0.0, 0.0, 0.9987502603949663),
         (0, 0, 0.5, 0, 0, 0, 1),
         (0, 0, 0.5, 0.1, 0.2, 0.3, 0.9273618495495703)]


def update_section(rotation: str) -> dict:
    m = oracle.OracleMap(21.12, 0.33)
    n = m.rows
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m.set_layer("ground", ((i + j * n).astype(np.float32) * F32(0.125)).astype(np.float32))
    m.set_layer("groundpatch", (F32(0.5) + ((i * 31 + j * 17) % 64).astype(np.float32) / F32(256.0)).astype(np.float32))
    steps = []
    for mv, pose in zip(MOVES, POSES):
        moved, _ = m.update(mv[0], mv[1], pose, rotation=rotation)
        steps.append({"moved": int(moved), "position": [hex64(m.position[0]), hex64(m.position[1])],
                      "ground": [hex32(v) for v in m.layer("ground").reshape(-1, order="F")],
                      "groundpatch": [hex32(v) for v in m.layer("groundpatch").reshape(-1, order="F")]})
    return {"update": steps}


def transform_section(rotation: str) -> dict:
    g = Lcg(0x7F200004)
    vals = []
    for k in range(600):
        q = [g.unit() - 0.5, g.unit() - 0.5, g.unit() - 0.5, g.unit() - 0.5]
        if k % 4:
            nrm = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
            q = [v / nrm for v in q]
        t = [(g.unit() - 0.5) * 2000.0, (g.unit() - 0.5) * 2000.0, (g.unit() - 0.5) * 20.0]
        p = [float(F32((g.unit() - 0.5) * 160.0)), float(F32((g.unit() - 0.5) * 160.0)), float(F32((g.unit() - 0.5) * 10.0))]
        R = oracle.rotation_from_quaternion(q, rotation)
        for r in range(3):  # dot product left to right, then + origin (tf2::Transform and KDL::Frame alike)
            vals.append(hex64(((float(R[r, 0]) * p[0] + float(R[r, 1]) * p[1]) + float(R[r, 2]) * p[2]) + t[r]))
    return {"do_transform": vals}


def libm_section() -> dict:
    g = Lcg(0x11B30005)
    L = oracle.lib()
    hyp = []
    for _ in range(4000):
        x = F32(g.wide_float() * F32(0.01))
        y = F32(g.wide_float() * F32(0.01))
        hyp.append(hex32(L.ggo_hypotf(C.c_float(float(x)), C.c_float(float(y)))))
    m = oracle.OracleMap(120.0, 0.33)  # (kept alive: expected_points() is a view of the map's own memory)
    e = m.expected_points().copy()
    exp = [hex32(e[i, j]) for i in range(0, 364, 7) for j in range(0, 364, 11)]
    return {"hypotf": hyp, "expected_points": exp}


def gridmap_message_bytes(rows, cols, resolution, length, position, layers, stamp, frame_id="map", seq=0, basic_layers=()) -> bytes:
    """grid_map_msgs/GridMap in ROS 1 wire format the way libgroundgrid_hip's gg_get_gridmap_message lays it out (restated here in
    Python so that the probe's bytes can be held against it without a GPU): info {header, resolution, length, pose}, layers,
    basic_layers, one Float32MultiArray per layer (dims column_index / row_index, column-major data), start indices 0."""
    out = bytearray()
    u32 = lambda v: out.extend(struct.pack("<I", v))
    f64 = lambda v: out.extend(struct.pack("<d", v))

    def string(t):
        u32(len(t))
        out.extend(t.encode())

    u32(seq), u32(stamp[0]), u32(stamp[1]), string(frame_id)
    f64(resolution), f64(length[0]), f64(length[1])
    f64(position[0]), f64(position[1]), f64(0.0)
    f64(0.0), f64(0.0), f64(0.0), f64(1.0)
    u32(len(layers))
    for name, _ in layers:
        string(name)
    u32(len(basic_layers))
    for name in basic_layers:
        string(name)
    u32(len(layers))
    for _, plane in layers:
        u32(2)
        string("column_index"), u32(cols), u32(rows * cols)
        string("row_index"), u32(rows), u32(rows)
        u32(0)
        u32(rows * cols)
        out.extend(np.asarray(plane, dtype="<f4").reshape(-1, order="F").tobytes())
    out.extend(struct.pack("<HH", 0, 0))
    return bytes(out)


def gridmap_section() -> dict:
    """Section 6 of probe.cpp (only when it was built with -DPIN_WITH_GRID_MAP_ROS): a 5 x 4 map of two layers."""
    res = 0.33  # (the probe passes a double; setGeometry keeps it, size = round(length / resolution), length = size * resolution)
    rows, cols = int(round(1.65 / res)), int(round(1.32 / res))
    g = Lcg(0x6D5A0006)
    layers = []
    for name in ("points", "ground"):
        plane = np.zeros((rows, cols), dtype=np.float32)
        for j in range(cols):
            for i in range(rows):
                plane[i, j] = g.wide_float()
        layers.append((name, plane))
    msg = gridmap_message_bytes(rows, cols, res, (rows * res, cols * res), (0.5, -0.25), layers, (1234, 5678))
    return {"gridmap_msg": list(msg)}


def emit(eigen: int, rotation: str) -> dict:
    doc = {"format": 1, "eigen_version": [3, 4 if eigen else 3, 0], "eigen_packet_floats": 4, "eigen_avx": 0, "emitted_by": "oracle"}
    doc.update(eigen_section(eigen))
    doc.update(grid_section())
    doc.update(update_section(rotation))
    doc.update(transform_section(rotation))
    doc.update(libm_section())
    doc.update(gridmap_section())
    return doc


def same(a, b) -> bool:
    return list(a) == list(b)


def compare(doc: dict) -> int:
    report, failed = [], False

    def section(name, variants, keys, chooser=None):
        nonlocal failed
        hits = []
        for label, mine in variants:
            ok = all(same(doc[k], mine[k]) for k in keys if k in doc)
            missing = [k for k in keys if k not in doc]
            if ok and not missing:
                hits.append(label)
        if hits:
            report.append(f"PASS  {name}: reproduced by {' / '.join(hits)}")
        else:
            failed = True
            detail = []
            for label, mine in variants:
                bad = [k for k in keys if k in doc and not same(doc[k], mine[k])]
                for k in bad[:3]:
                    n_bad = sum(1 for x, y in zip(doc[k], mine[k]) if x != y) + abs(len(doc[k]) - len(mine[k]))
                    detail.append(f"{label}:{k} {n_bad}/{len(doc[k])} differ")
            report.append(f"FAIL  {name}: no oracle variant matches ({'; '.join(detail)})")
        return hits

    print(f"probe: Eigen {'.'.join(map(str, doc.get('eigen_version', ['?'])))}, packet = {doc.get('eigen_packet_floats')} floats, "
          f"AVX = {doc.get('eigen_avx')}")
    e = [("GG_EIGEN_33 (eigen_reduction=0)", eigen_section(0)), ("GG_EIGEN_34_SSE (eigen_reduction=1)", eigen_section(1))]
    section("Eigen 3x3 block sums / products (:268, :457-458)", e, ["eigen_sum3", "eigen_prod3", "eigen_arrprod3", "eigen_min3"])
    eig = section("Eigen 5x5 block sums / products (:359, :374-375)", e, ["eigen_sum5", "eigen_prod5", "eigen_arrprod5", "eigen_min5"])
    section("Eigen element-wise variance (:323)", e[:1], ["eigen_variance"])
    gsec = grid_section()
    section("grid_map setGeometry (GroundGrid.cpp:58)", [("oracle", gsec)], ["gm_size", "gm_geometry"])
    section("grid_map getIndex / isInside on cell edges (:228-230, :261)", [("oracle", gsec)], ["gm_index", "gm_index_random"])

    def update_flat(d):
        out = {"update_moved": [s["moved"] for s in d["update"]], "update_position": [p for s in d["update"] for p in s["position"]]}
        out["update_ground"] = [v for s in d["update"] for v in s["ground"]]
        out["update_groundpatch"] = [v for s in d["update"] for v in s["groundpatch"]]
        return out

    if "update" in doc:
        doc.update(update_flat(doc))
    rot = section("GroundGrid::update: move + exposed-cell plane + re-linearisation (GroundGrid.cpp:83-147)",
                  [(f'rotation="{r}"', update_flat(update_section(r))) for r in ("kdl", "tf2")],
                  ["update_moved", "update_position", "update_ground", "update_groundpatch"])
    rot2 = section("tf2::doTransform(PointStamped) (GroundGridNodelet.cpp:146,176)",
                   [(f'rotation="{r}"', transform_section(r)) for r in ("kdl", "tf2")], ["do_transform"])
    section("glibc hypotf / atanf (:170, :44)", [("oracle", libm_section())], ["hypotf", "expected_points"])
    if "gridmap_msg" in doc:  # (optional: the probe was built with -DPIN_WITH_GRID_MAP_ROS)
        section("grid_map_msgs/GridMap wire bytes (GroundGridNodelet.cpp:211-214; gg_get_gridmap_message)", [("library layout", gridmap_section())], ["gridmap_msg"])
    else:
        report.append("SKIP  grid_map_msgs/GridMap wire bytes: the probe was built without -DPIN_WITH_GRID_MAP_ROS")
    print("\n".join(report))
    if not failed:
        print("\nThe oracle is pinned by these vectors.  Select:")
        print(f"  eigen: {eig[0]}")
        both = [r for r in rot if r in rot2]
        print(f"  rotation: {both[0] if both else (rot + rot2)[0]}  (api.GridMap.move(..., rotation=), kitti.ROTATION_CONVENTION, gg_transform_from_pose)")
    return 1 if failed else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("vectors", nargs="?")
    ap.add_argument("--emit")
    ap.add_argument("--eigen", type=int, default=0)
    ap.add_argument("--rotation", default="kdl")
    a = ap.parse_args()
    if a.emit:
        json.dump(emit(a.eigen, a.rotation), open(a.emit, "w"))
        return 0
    if not a.vectors:
        ap.error("vectors.json (from ./probe) or --emit")
    return compare(json.load(open(a.vectors)))


if __name__ == "__main__":
    sys.exit(main())
