"""SemanticKITTI sequence reader and the frame wiring of the reference's evaluation pipeline, without ROS.

Restates what /root/reference/scripts/kitti_data_publisher.py and launch/KITTIEvaluate.launch feed the nodelet:

* file formats (kitti_data_publisher.py:117-161): ``velodyne/%06d.bin`` float32 x,y,z,remission; ``labels/%06d.label``
  uint32 whose low 16 bits are the semantic label -- the player stores it in the cloud's ``ring`` field (:124,:130);
  ``poses.txt`` 3x4 row-major per line; ``times.txt``;
* pose math (:164-180): pose_i = calib^-1 . P_i . calib with the hard-coded calibration string (:168);
* frames: the player broadcasts map <- kitti_base_link = pose_i (:207-215) and publishes the odometry position
  = pose translation (:193-195); static transforms (KITTIEvaluate.launch:13-16): base_link = kitti_base_link +
  (1.95, 0, -1.73), velodyne = kitti_base_link;
* what the nodelet derives per cloud (src/GroundGridNodelet.cpp:129-195): the cloud transformed point by point into the
  map frame in double precision and cast back to float (:166-181), the sensor origin = map <- velodyne applied to 0,
  mapToBase = map <- base_link (only translation.z is used by the path), and in odom_callback
  (src/GroundGrid.cpp:83-147) the base_link <- map transform used to seed newly exposed cells.

tf timing / interpolation effects of the live ROS pipeline are not modelled: every cloud uses its own pose exactly, so
aggregate results are a close match to README.md:57-94, not a bit-exact one.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Iterator, List, Optional

import numpy as np

from .synth import POINT_DTYPE, clone_cloud, empty_cloud

# scripts/kitti_data_publisher.py:168
CALIB_STRING = ("4.276802385584e-04 -9.999672484946e-01 -8.084491683471e-03 -1.198459927713e-02 -7.210626507497e-03 "
                "8.081198471645e-03 -9.999413164504e-01 -5.403984729748e-02 9.999738645903e-01 4.859485810390e-04 "
                "-7.206933692422e-03 -2.921968648686e-01")
# launch/KITTIEvaluate.launch:13,16
KITTI_BASE_TO_BASE_LINK = (1.95, 0.0, -1.73)
KITTI_BASE_TO_VELODYNE = (0.0, 0.0, 0.0)


def read_bin(path: str) -> np.ndarray:
    """velodyne/%06d.bin -> (n, 4) float32 x, y, z, remission (kitti_data_publisher.py:120-121)."""
    return np.fromfile(path, dtype=np.float32).reshape((-1, 4))


def read_labels(path: str) -> np.ndarray:
    """labels/%06d.label -> semantic label = low 16 bits (kitti_data_publisher.py:156-161)."""
    return (np.fromfile(path, dtype=np.uint32).reshape(-1) & 0xFFFF).astype(np.uint16)


def make_cloud(scan: np.ndarray, labels: Optional[np.ndarray]) -> np.ndarray:
    """PointXYZIR records as the nodelet receives them: intensity = remission, ring = semantic label (:124-130)."""
    pts = empty_cloud(scan.shape[0])
    pts["x"], pts["y"], pts["z"], pts["intensity"] = scan[:, 0], scan[:, 1], scan[:, 2], scan[:, 3]
    if labels is not None:
        pts["ring"] = labels
    return pts


def read_poses(path: str) -> List[np.ndarray]:
    """poses.txt -> list of 4x4 float64 calib^-1 . P . calib (kitti_data_publisher.py:164-180)."""
    calib = np.vstack((np.array(CALIB_STRING.split(), dtype=np.float64).reshape(3, 4), [0, 0, 0, 1]))
    calib_inv = np.linalg.inv(calib)
    poses = []
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            P = np.vstack((np.array(line.split(), dtype=np.float64).reshape(3, 4), [0, 0, 0, 1]))
            poses.append(np.matmul(calib_inv, np.matmul(P, calib)))
    return poses


def quaternion_from_matrix(M: np.ndarray) -> np.ndarray:
    """tf.transformations.quaternion_from_matrix (x, y, z, w), the branchy trace form used by ROS Noetic's tf."""
    M = np.asarray(M, dtype=np.float64)[:4, :4]
    q = np.empty(4)
    t = np.trace(M)
    if t > M[3, 3]:
        q[3] = t
        q[2] = M[1, 0] - M[0, 1]
        q[1] = M[0, 2] - M[2, 0]
        q[0] = M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i] = t
        q[j] = M[i, j] + M[j, i]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
    q *= 0.5 / np.sqrt(t * M[3, 3])
    return q


def euler_roundtrip_quaternion(q) -> np.ndarray:
    """What the player actually broadcasts as map <- kitti_base_link (scripts/kitti_data_publisher.py:201-214): not the pose's
    quaternion q but quaternion_from_euler(*euler_from_quaternion(q_new)) with q_new = quaternion_multiply((0, 0, 0, 1), q) --
    a round trip through roll / pitch / yaw ("sxyz") in tf.transformations (restated from transformations.py as shipped with
    ROS Noetic's tf; third-party, unpinned like the other conventions).  The result differs from q in the last ulps."""
    import math

    x0, y0, z0, w0 = (float(v) for v in q)
    x1, y1, z1, w1 = 0.0, 0.0, 0.0, 1.0  # quaternion_from_euler(0, 0, 0)
    qn = np.array((x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0, -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                   x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0, -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0), dtype=np.float64)
    eps = np.finfo(float).eps * 4.0
    # euler_from_quaternion = euler_from_matrix(quaternion_matrix(q), 'sxyz')
    qq = np.array(qn, dtype=np.float64, copy=True)
    nq = np.dot(qq, qq)
    if nq < eps:
        M = np.identity(3)
    else:
        qq *= math.sqrt(2.0 / nq)
        o = np.outer(qq, qq)
        M = np.array(((1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]),
                      (o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]),
                      (o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1])), dtype=np.float64)
    cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
    if cy > eps:
        ax, ay, az = math.atan2(M[2, 1], M[2, 2]), math.atan2(-M[2, 0], cy), math.atan2(M[1, 0], M[0, 0])
    else:
        ax, ay, az = math.atan2(-M[1, 2], M[1, 1]), math.atan2(-M[2, 0], cy), 0.0
    # quaternion_from_euler(ax, ay, az, 'sxyz')
    ai, aj, ak = ax / 2.0, ay / 2.0, az / 2.0
    ci, si, cj, sj, ck, sk = math.cos(ai), math.sin(ai), math.cos(aj), math.sin(aj), math.cos(ak), math.sin(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array((cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc, cj * cc + sj * ss), dtype=np.float64)


# Which rotation matrix tf2::doTransform(PointStamped) -- the overload the reference calls for the cloud transform, the
# cloud origin and GroundGrid::update (src/GroundGridNodelet.cpp:146,176, src/GroundGrid.cpp:129) -- builds from the message
# quaternion is a third-party convention (tools/pin/ decides it on a ROS box): ROS Melodic / Noetic route that overload
# through KDL (gmTransformToKDL -> KDL::Rotation::Quaternion), the Point / Vector3 overloads through tf2::Matrix3x3.
ROTATION_CONVENTION = "kdl"


def matrix_from_quaternion(q, rotation: str = None) -> np.ndarray:
    """Quaternion (x, y, z, w) -> rotation matrix, "tf2" = tf2::Matrix3x3::setRotation, "kdl" = KDL::Rotation::Quaternion
    (same operation order as the C helpers gg_rotation_from_quaternion / ggo_rotation_from_quaternion; Python floats are
    IEEE doubles, so the entries are bit-identical to theirs)."""
    rotation = rotation or ROTATION_CONVENTION
    x, y, z, w = (float(v) for v in q)
    if rotation == "tf2":
        d = x * x + y * y + z * z + w * w
        s = 2.0 / d
        xs, ys, zs = x * s, y * s, z * s
        wx, wy, wz = w * xs, w * ys, w * zs
        xx, xy, xz = x * xs, x * ys, x * zs
        yy, yz, zz = y * ys, y * zs, z * zs
        return np.array([[1.0 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1.0 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1.0 - (xx + yy)]])
    assert rotation == "kdl", rotation
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    return np.array([[w2 + x2 - y2 - z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
                     [2 * x * y + 2 * w * z, w2 - x2 + y2 - z2, 2 * y * z - 2 * w * x],
                     [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, w2 - x2 - y2 + z2]])


def transform_cloud(cloud: np.ndarray, R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """src/GroundGridNodelet.cpp:166-181: tf2::doTransform per point in double (dot products left to right,
    then + origin), cast back to float; everything else of the record is copied."""
    x, y, z = (cloud[k].astype(np.float64) for k in ("x", "y", "z"))
    out = clone_cloud(cloud)
    for name, row, off in (("x", R[0], t[0]), ("y", R[1], t[1]), ("z", R[2], t[2])):
        out[name] = (((row[0] * x + row[1] * y) + row[2] * z) + off).astype(np.float32)
    return out


@dataclass
class Frame:
    index: int
    cloud_sensor: np.ndarray      # PointXYZIR in the kitti_base_link (= velodyne) frame, ring = semantic label
    cloud_map: np.ndarray         # the same cloud in the map frame (what filter_cloud receives)
    origin: tuple                 # cloudOrigin: map <- velodyne applied to (0, 0, 0), as floats
    odom: tuple                   # odometry position x, y, z (GroundGrid::update / initGroundGrid)
    map_to_base_z: float          # mapToBase.transform.translation.z
    base_to_map: tuple            # (tx, ty, tz, qx, qy, qz, qw) of base_link <- map


class KittiSequence:
    """Iterates a SemanticKITTI sequence directory (``.../sequences/00``) in the reference pipeline's conventions."""

    def __init__(self, directory: str, euler_roundtrip: bool = False):
        self.dir = directory
        self.euler_roundtrip = euler_roundtrip  # model the player's quaternion -> euler -> quaternion round trip (:208-214)
        self.poses = read_poses(os.path.join(directory, "poses.txt"))
        self.n = len(self.poses)
        self.have_labels = os.path.isdir(os.path.join(directory, "labels"))

    def __len__(self):
        return self.n

    def frame(self, i: int) -> Frame:
        scan = read_bin(os.path.join(self.dir, "velodyne", f"{i:06d}.bin"))
        labels = read_labels(os.path.join(self.dir, "labels", f"{i:06d}.label")) if self.have_labels else None
        return make_frame(i, make_cloud(scan, labels), self.poses[i], euler_roundtrip=self.euler_roundtrip)

    def __iter__(self) -> Iterator[Frame]:
        for i in range(self.n):
            yield self.frame(i)


def make_frame(i: int, cloud_sensor: np.ndarray, pose: np.ndarray, euler_roundtrip: bool = False) -> Frame:
    q = quaternion_from_matrix(pose)                  # player: quaternion of the pose (:199)
    if euler_roundtrip:                               # ... which it broadcasts after a trip through euler angles (:208-214)
        q = euler_roundtrip_quaternion(q)
    R = matrix_from_quaternion(q)                     # nodelet: doTransform rebuilds the rotation from the quaternion
    t = np.array([pose[0, 3], pose[1, 3], pose[2, 3]])
    cloud_map = transform_cloud(cloud_sensor, R, t)
    origin = tuple(np.float32(v) for v in t)          # Nodelet.cpp:139-146,192-195 (velodyne == kitti_base_link)
    # map <- base_link = pose . static(1.95, 0, -1.73)
    sb = np.array(KITTI_BASE_TO_BASE_LINK)
    base_in_map = R @ sb + t
    # base_link <- map = inverse: rotation conj(q), translation -R^T . base_in_map
    tb = -(R.T @ base_in_map)
    base_to_map = (tb[0], tb[1], tb[2], -q[0], -q[1], -q[2], q[3])
    return Frame(i, cloud_sensor, cloud_map, origin, (t[0], t[1], t[2]), float(base_in_map[2]), base_to_map)


def write_synthetic_sequence(directory: str, clouds: List[np.ndarray], poses_cam: List[np.ndarray]):
    """Writes clouds (POINT_DTYPE, ring = label) + camera-frame 3x4 poses in SemanticKITTI layout (tests / demos)."""
    os.makedirs(os.path.join(directory, "velodyne"), exist_ok=True)
    os.makedirs(os.path.join(directory, "labels"), exist_ok=True)
    with open(os.path.join(directory, "poses.txt"), "w") as fp, open(os.path.join(directory, "times.txt"), "w") as ft:
        for i, (c, P) in enumerate(zip(clouds, poses_cam)):
            np.column_stack([c["x"], c["y"], c["z"], c["intensity"]]).astype(np.float32).tofile(os.path.join(directory, "velodyne", f"{i:06d}.bin"))
            c["ring"].astype(np.uint32).tofile(os.path.join(directory, "labels", f"{i:06d}.label"))
            fp.write(" ".join(f"{v:.12e}" for v in np.asarray(P, dtype=np.float64)[:3, :4].reshape(-1)) + "\n")
            ft.write(f"{0.1 * i:.6e}\n")


def drive_poses(n_frames: int, metres_per_frame: float = 0.8) -> List[np.ndarray]:
    """Vehicle poses (4 x 4, map <- kitti_base_link) of a closed loop driven once in n_frames steps: a circle with a slow
    lateral weave, the heading along the path and a gentle climb and descent -- sequence 00's shape (3.7 km in 4540 frames,
    README.md:57) without its data: the map scrolls two or three cells in every frame."""
    radius = n_frames * metres_per_frame / (2.0 * np.pi)
    poses = []
    for i in range(n_frames):
        th = 2.0 * np.pi * i / n_frames
        rad = radius * (1.0 + 0.01 * np.sin(9.0 * th))
        x, y = rad * np.cos(th) - radius, rad * np.sin(th)
        yaw = th + np.pi / 2.0 + 0.02 * np.sin(31.0 * th)
        T = np.eye(4)
        T[:3, :3] = [[np.cos(yaw), -np.sin(yaw), 0.0], [np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, 1.0]]
        T[:3, 3] = [x, y, 1.5 * np.sin(3.0 * th)]
        poses.append(T)
    return poses


def synthetic_drive(n_frames: int = 4540, n_scenes: int = 8, n_az: int = 2083, seed: int = 20240113, euler_roundtrip: bool = False) -> Iterator[Frame]:
    """BASELINE configs[4]'s SHAPE on synthetic data, in memory: n_frames consecutive full-size HDL-64E clouds (n_scenes seeded
    scenes of groundgrid_amd.synth in turn, semantic labels faked from the height: `road` below, `building` above, some
    `vegetation`) along drive_poses(), wired exactly like a sequence directory (make_frame).  The dataset itself is not in the
    image; what this exercises is what the dataset would: thousands of dependent frames, the map scrolling in every one."""
    from .synth import hdl64_cloud

    scenes = []
    for k in range(n_scenes):
        c = hdl64_cloud(seed=seed + k, n_az=n_az)
        lab = np.where(c["z"] < -1.4, 40, 50).astype(np.uint16)
        lab[::17] = 70
        c["ring"] = lab
        scenes.append(c)
    for i, pose in enumerate(drive_poses(n_frames)):
        yield make_frame(i, scenes[i % n_scenes], pose, euler_roundtrip=euler_roundtrip)
