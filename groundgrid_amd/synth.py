"""Synthetic LiDAR clouds for the benchmark / parity configurations.

There is no dataset on the build or GPU boxes, so BASELINE.json's configs 2-4 are driven by a
seeded ray caster:  a spinning multi-beam sensor at the map origin over an analytic terrain
``z = -1.73 + 0.02 x + 0.3 sin(x/15) cos(y/20)`` with axis-aligned boxes (cars, walls), first
hit wins, Gaussian range noise.  Output records use the reference's 32-byte
``velodyne_pointcloud::PointXYZIR`` layout (include/velodyne_pointcloud/point_types.h:27-33).
"""
from __future__ import annotations

import numpy as np

# 32-byte PointXYZIR record: x,y,z @0,4,8 ; intensity @16 ; ring(u16) @20
POINT_DTYPE = np.dtype(
    {
        "names": ["x", "y", "z", "intensity", "ring"],
        "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"],
        "offsets": [0, 4, 8, 16, 20],
        "itemsize": 32,
    }
)

SENSOR_HEIGHT = 1.73  # launch/KITTIEvaluate.launch:13 (base_link is 1.73 m below the sensor)


def empty_cloud(n: int) -> np.ndarray:
    return np.zeros(n * 32, dtype=np.uint8).view(POINT_DTYPE)


def clone_cloud(cloud: np.ndarray) -> np.ndarray:
    """Byte-exact copy.  (ndarray.copy() of a padded structured dtype copies field by field and leaves the 14 padding
    bytes of each record undefined; the reference copies whole 32-byte points.)"""
    return np.frombuffer(cloud.tobytes(), dtype=np.uint8).copy().view(POINT_DTYPE)


def make_cloud(xyz, ring=None, intensity=None) -> np.ndarray:
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    pts = empty_cloud(xyz.shape[0])
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if ring is not None:
        pts["ring"] = ring
    if intensity is not None:
        pts["intensity"] = intensity
    return pts


def hdl64_elevations_deg() -> np.ndarray:
    """HDL-64E S2: 32 upper lasers +2.0 .. -8.33 deg (1/3 deg), 32 lower -8.83 .. -24.33 deg (1/2 deg)."""
    upper = 2.0 - np.arange(32) / 3.0
    lower = -8.83 - np.arange(32) * 0.5
    return np.concatenate([upper, lower])


def _terrain(x, y, scale=1.0):
    return -SENSOR_HEIGHT + 0.02 * x + 0.3 * np.sin(x / (15.0 * scale)) * np.cos(y / (20.0 * scale))


def _ground_hit(dx, dy, dz, max_range, scale):
    """First intersection range of rays (origin 0) with the terrain; inf if none within max_range."""
    step = 0.5
    nstep = int(max_range / step) + 1
    t_lo = np.full(dx.shape, np.inf, dtype=np.float32)
    t_hi = np.full(dx.shape, np.inf, dtype=np.float32)
    found = np.zeros(dx.shape, dtype=bool)
    prev_above = np.ones(dx.shape, dtype=bool)  # the sensor is above the terrain
    for k in range(1, nstep + 1):
        t = np.float32(k * step)
        above = (t * dz) > _terrain(t * dx, t * dy, scale)
        newly = prev_above & ~above & ~found
        if newly.any():
            t_lo[newly] = t - step
            t_hi[newly] = t
            found |= newly
        prev_above = above
        if found.all():
            break
    idx = np.nonzero(found)[0]
    lo, hi = t_lo[idx].astype(np.float64), t_hi[idx].astype(np.float64)
    ddx, ddy, ddz = dx[idx].astype(np.float64), dy[idx].astype(np.float64), dz[idx].astype(np.float64)
    for _ in range(12):  # bisection to ~0.1 mm
        mid = 0.5 * (lo + hi)
        above = (mid * ddz) > _terrain(mid * ddx, mid * ddy, scale)
        lo = np.where(above, mid, lo)
        hi = np.where(above, hi, mid)
    out = np.full(dx.shape, np.inf, dtype=np.float64)
    out[idx] = 0.5 * (lo + hi)
    return out


def _box_hits(dx, dy, dz, boxes):
    """Slab test of rays (origin 0) against axis-aligned boxes [(xmin,xmax,ymin,ymax,zmin,zmax)]."""
    best = np.full(dx.shape, np.inf, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = [1.0 / dx.astype(np.float64), 1.0 / dy.astype(np.float64), 1.0 / dz.astype(np.float64)]
    for b in boxes:
        tmin = np.full(dx.shape, -np.inf)
        tmax = np.full(dx.shape, np.inf)
        for a in range(3):
            t1 = b[2 * a] * inv[a]
            t2 = b[2 * a + 1] * inv[a]
            tmin = np.maximum(tmin, np.minimum(t1, t2))
            tmax = np.minimum(tmax, np.maximum(t1, t2))
        hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
        best = np.where(hit & (tmin < best), tmin, best)
    return best


def _scene_boxes(rng, scale, n_boxes=40, enclosure=None):
    boxes = []
    for k in range(n_boxes):
        while True:
            cx, cy = rng.uniform(-55.0 * scale, 55.0 * scale, size=2)
            if np.hypot(cx, cy) > 6.0:
                break
        if k % 2 == 0:  # car 4 x 1.8 x 1.5
            sx, sy, sz = 4.0, 1.8, 1.5
        else:  # wall 20 x 0.3 x 3
            sx, sy, sz = 20.0, 0.3, 3.0
        if rng.random() < 0.5:
            sx, sy = sy, sx
        z0 = float(_terrain(cx, cy, scale)) - 0.2
        boxes.append((cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2, z0, z0 + 0.2 + sz))
    if enclosure is not None:  # four tall walls: an "urban canyon" so upward beams return too
        r, h, th = enclosure
        boxes += [
            (r, r + th, -r - th, r + th, -10.0, h),
            (-r - th, -r, -r - th, r + th, -10.0, h),
            (-r - th, r + th, r, r + th, -10.0, h),
            (-r - th, r + th, -r - th, -r, -10.0, h),
        ]
    return boxes


def _cast(elev_deg, n_az, rng, max_range, min_range, scale, boxes, order, noise):
    n_ring = elev_deg.shape[0]
    el = np.deg2rad(elev_deg).astype(np.float32)
    az = (np.arange(n_az, dtype=np.float32) * np.float32(2.0 * np.pi / n_az)).astype(np.float32)
    if order == "ring":  # KITTI-like: one ring after the other
        EL, AZ = np.meshgrid(el, az, indexing="ij")
        RING = np.repeat(np.arange(n_ring, dtype=np.uint16), n_az)
    elif order == "azimuth":  # firing order: all lasers of one azimuth column together
        AZ, EL = np.meshgrid(az, el, indexing="ij")
        RING = np.tile(np.arange(n_ring, dtype=np.uint16), n_az)
    else:
        raise ValueError(order)
    EL, AZ = EL.ravel(), AZ.ravel()
    dx = (np.cos(EL) * np.cos(AZ)).astype(np.float32)
    dy = (np.cos(EL) * np.sin(AZ)).astype(np.float32)
    dz = np.sin(EL).astype(np.float32)
    t = np.minimum(_ground_hit(dx, dy, dz, max_range, scale), _box_hits(dx, dy, dz, boxes))
    t = t + rng.normal(0.0, noise, size=t.shape)
    keep = np.isfinite(t) & (t >= min_range) & (t <= max_range)
    t = t[keep]
    xyz = np.stack([t * dx[keep], t * dy[keep], t * dz[keep]], axis=1).astype(np.float32)
    inten = rng.uniform(0.0, 1.0, size=t.shape).astype(np.float32)
    return make_cloud(xyz, ring=RING[keep], intensity=inten)


def hdl64_cloud(seed: int = 20240113, order: str = "ring", n_az: int = 2083) -> np.ndarray:
    """BASELINE config 2: Velodyne HDL-64E, 64 x 2083 rays, range 2.5..80 m  ->  ~120 k points."""
    rng = np.random.Generator(np.random.PCG64(seed))
    boxes = _scene_boxes(rng, 1.0)
    return _cast(hdl64_elevations_deg(), n_az, rng, 80.0, 2.5, 1.0, boxes, order, 0.02)


def os128_cloud(seed: int = 20240113, order: str = "ring", n_az: int = 16384, n_ring: int = 128) -> np.ndarray:
    """BASELINE config 4: dense 128-beam (+-22.5 deg) x 16384 azimuths, range <= 100 m, scene x1.7,
    enclosed by 45 m walls at +-95 m so that (nearly) every beam returns  ->  ~2.1 M points."""
    rng = np.random.Generator(np.random.PCG64(seed))
    boxes = _scene_boxes(rng, 1.7, enclosure=(95.0 / np.sqrt(2.0) - 1.0, 45.0, 1.0))
    elev = np.linspace(22.5, -22.5, n_ring)
    return _cast(elev, n_az, rng, 100.0, 2.5, 1.7, boxes, order, 0.02)


def os128_cloud_fast(seed: int = 20240113) -> np.ndarray:
    """The same workload shape as os128_cloud (128 rings x 16384 azimuths, ~2.1 M returns, ring-major order) in a quarter of
    the ray-casting time: 4096 azimuths are cast and the cloud is completed with three copies rotated by 1/4, 2/4, 3/4 of that
    azimuth step (the scene rotates along by at most 0.066 degrees: a synthetic scene does not mind).  For benchmarks."""
    base = os128_cloud(seed=seed, n_az=4096)
    braw = np.ascontiguousarray(base).view(np.uint8).reshape(-1, 32)
    parts = []
    for k in range(4):
        ang = np.float32(k * 2.0 * np.pi / 16384.0)
        c, s = np.cos(ang), np.sin(ang)
        praw = braw.copy()
        f = praw.view(np.float32).reshape(-1, 8)  # x, y, z, pad, intensity, ...
        bx, by = base["x"], base["y"]
        f[:, 0] = (c * bx - s * by).astype(np.float32)
        f[:, 1] = (s * bx + c * by).astype(np.float32)
        parts.append(praw)
    raw = np.concatenate(parts, axis=0)  # whole 32-byte records
    out = raw.reshape(-1).view(POINT_DTYPE)
    az = np.arctan2(out["y"].astype(np.float64), out["x"].astype(np.float64)) % (2.0 * np.pi)
    order = np.lexsort((az, out["ring"]))  # ring after ring, azimuth increasing inside a ring
    return np.ascontiguousarray(raw[order]).reshape(-1).view(POINT_DTYPE)


def random_cloud(n: int, seed: int = 0, extent: float = 70.0, max_ring: int = 64) -> np.ndarray:
    """Unstructured stress cloud: uniform xy (partly outside a 120 m map), noisy terrain + clutter."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xy = rng.uniform(-extent, extent, size=(n, 2))
    z = _terrain(xy[:, 0], xy[:, 1]) + rng.normal(0.0, 0.03, size=n)
    clutter = rng.random(n) < 0.25
    z = np.where(clutter, z + rng.uniform(0.0, 3.0, size=n), z)
    below = rng.random(n) < 0.02
    z = np.where(below, z - rng.uniform(0.3, 2.0, size=n), z)
    xyz = np.column_stack([xy, z]).astype(np.float32)
    return make_cloud(xyz, ring=rng.integers(0, max_ring, size=n).astype(np.uint16),
                      intensity=rng.uniform(0, 1, size=n).astype(np.float32))
