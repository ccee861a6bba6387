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
