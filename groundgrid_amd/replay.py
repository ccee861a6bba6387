"""Sequence replay (BASELINE.json configs[0] / configs[4]): the call sequence of the reference nodelet for a stream of
clouds -- odom_callback -> GroundGrid::update, points_callback -> filter_cloud, evaluator -- without ROS
(/root/reference/src/GroundGridNodelet.cpp:107-232, scripts/eval_groundpoint_classifier.py:95-132).

`backend` is anything with reset(pos_xy, odom_z), move(odom_xy, base_to_map) and filter(cloud_map, origin, base_z) ->
(labels per input point, index per input point); DeviceBackend is the MI355X one.
"""
from __future__ import annotations

import time

import numpy as np

from .evaluate import GroundEvaluator


class DeviceBackend:
    def __init__(self, max_points: int = 200_000, dimension: float = 120.0, resolution: float = 0.33, device: int = 0):
        from . import api

        self.seg = api.GroundSegmentation().init(dimension, resolution, n_slots=1, max_points=max_points, device=device)
        self.map = self.seg.map(0)

    def reset(self, pos, odom_z):
        self.map.reset(odom_z=odom_z, pos=pos)

    def move(self, odom, base_to_map):
        from . import kitti

        self.map.move(odom[0], odom[1], base_to_map, rotation=kitti.ROTATION_CONVENTION)

    def filter(self, cloud_map, origin, base_z):
        _, labels, index = self.seg.filter_cloud(cloud_map, origin, base_z, return_details=True)
        return labels, index


def replay(frames, backend, evaluator: GroundEvaluator = None, on_frame=None):
    """frames: iterable of groundgrid_amd.kitti.Frame.  Returns (evaluator, seconds spent inside the backend)."""
    ev = evaluator or GroundEvaluator()
    spent = 0.0
    first = True
    for fr in frames:
        t0 = time.perf_counter()
        if first:
            backend.reset((fr.odom[0], fr.odom[1]), np.float32(fr.odom[2]))  # GroundGrid::initGroundGrid (src/GroundGrid.cpp:50-80)
            first = False
        else:
            backend.move((fr.odom[0], fr.odom[1]), fr.base_to_map)            # GroundGrid::update
        labels, index = backend.filter(fr.cloud_map, fr.origin, fr.map_to_base_z)
        spent += time.perf_counter() - t0
        emitted = index >= 0                                                   # the evaluator only sees the returned cloud
        ev.add_cloud(labels[emitted], fr.cloud_map["ring"][emitted])
        if on_frame:
            on_frame(fr, labels, index)
    return ev, spent


def replay_side_by_side(frames, device, cpu, cpu_frames: int = 0):
    """One pass over `frames` through two backends (`cpu`: the checker -- tests and bench.py pass the oracle's; None: device only), frame by frame: (evaluator of the device path, its seconds, the CPU path's
    seconds and frames, labels_equal_in_every_frame, first frame that differed or -1).  cpu_frames > 0 stops the CPU path (and the
    comparison) after that many frames."""
    ev = GroundEvaluator()
    t_dev = t_cpu = 0.0
    n_cpu = 0
    first_bad = -1
    first = True
    for k, fr in enumerate(frames):
        with_cpu = cpu is not None and (cpu_frames <= 0 or k < cpu_frames)
        t0 = time.perf_counter()
        if first:
            device.reset((fr.odom[0], fr.odom[1]), np.float32(fr.odom[2]))
        else:
            device.move((fr.odom[0], fr.odom[1]), fr.base_to_map)
        labels, index = device.filter(fr.cloud_map, fr.origin, fr.map_to_base_z)
        t_dev += time.perf_counter() - t0
        if with_cpu:
            t0 = time.perf_counter()
            if first:
                cpu.reset((fr.odom[0], fr.odom[1]), np.float32(fr.odom[2]))
            else:
                cpu.move((fr.odom[0], fr.odom[1]), fr.base_to_map)
            lc, ic = cpu.filter(fr.cloud_map, fr.origin, fr.map_to_base_z)
            t_cpu += time.perf_counter() - t0
            n_cpu += 1
            if first_bad < 0 and not (np.array_equal(labels, lc) and np.array_equal(index, ic)):
                first_bad = k
        first = False
        emitted = index >= 0
        ev.add_cloud(labels[emitted], fr.cloud_map["ring"][emitted])
    return ev, t_dev, t_cpu, n_cpu, first_bad < 0, first_bad
