"""Second, independent restatement of the hot path in pure Python (numpy float32 / float64 scalars), for
SMALL cases only.  It is written directly from /root/reference/src/GroundSegmentation.cpp, separately from
oracle/gg_oracle.c, so that a transcription slip in either shows up as a disagreement
(tests/test_oracle_cpu.py).  Test infrastructure -- never imported by the product.

Every arithmetic step is an explicit np.float32 or np.float64 operation so that Python's own double
arithmetic never leaks in.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
f64 = np.float64
FLT_MIN = np.finfo(np.float32).tiny  # numeric_limits<float>::min()
FLT_MAX = np.finfo(np.float32).max


def tree_sum(e):
    """Eigen 3.3.7 redux_novec_unroller: f(s, n) = f(s, n//2) + f(s + n//2, n - n//2)."""
    n = len(e)
    if n == 1:
        return f32(e[0])
    h = n // 2
    return f32(tree_sum(e[:h]) + tree_sum(e[h:]))


EIGEN_REDUCTION = 0  # 0 = Eigen 3.3.x (every fixed-size block sum takes the unrolled tree), 1 = Eigen 3.4.x SSE2 build


def block_sum(e):
    """Block<MatrixXf,S,S>::sum() of the S*S column-major coefficients e.  Under Eigen 3.4 (Packet4f) the 5x5 blocks take
    the slice-vectorised reduction: four row lanes accumulated over the columns, predux (a0+a2)+(a1+a3), then row 4 of
    every column; 3x3 blocks keep the unrolled tree under both versions."""
    if EIGEN_REDUCTION == 0 or len(e) != 25:
        return tree_sum(e)
    lanes = [f32(e[r]) for r in range(4)]
    for j in range(1, 5):
        lanes = [f32(lanes[r] + f32(e[5 * j + r])) for r in range(4)]
    res = f32(f32(lanes[0] + lanes[2]) + f32(lanes[1] + lanes[3]))
    for j in range(5):
        res = f32(res + f32(e[5 * j + 4]))
    return res


def std_min(a, b):
    return b if b < a else a


def std_max(a, b):
    return b if a < b else a


def trunc_int(v):
    if not (-2147483649.0 < v < 2147483648.0):
        return -(2**31)
    return int(v)  # Python int() truncates toward zero


class PyRef:
    def __init__(self, length=120.0, resolution=0.33, pos=(0.0, 0.0), odom_z=0.0, cfg=None):
        res_f = f32(resolution)
        self.res = f64(res_f)
        n = int(round(float(f64(f32(length)) / self.res)))
        cell_count = int(np.round(f32(f32(int(f32(length))) / res_f)))  # size_t / float -> float; std::round
        assert n == cell_count
        self.n = n
        self.length = f64(n) * self.res
        self.pos = (f64(pos[0]), f64(pos[1]))
        self.vpad = f32(0.00174532925 * 2)
        self.min_dist_sq = f32(12.0)
        self.cfg = cfg or dict(
            point_count_cell_variance_threshold=10, max_ring=1024, distance_factor=0.0001,
            minimum_distance_factor=0.0005, miminum_point_height_threshold=0.3,
            minimum_point_height_obstacle_threshold=0.1, outlier_tolerance=0.1,
            ground_patch_detection_minimum_point_count_threshold=0.25, patch_size_change_distance=20.0,
            occupied_cells_decrease_factor=5.0, occupied_cells_point_count_factor=20.0,
            min_outlier_detection_ground_confidence=1.25,
        )
        z = lambda v: np.full((n, n), v, dtype=np.float32)  # noqa: E731
        self.L = dict(points=z(0), ground=z(f32(odom_z)), groundpatch=z(f32(0.0000001)), minGroundHeight=z(100.0),
                      maxGroundHeight=z(-100.0), groundCandidates=z(0), planeDist=z(0), m2=z(0), meanVariance=z(0),
                      pointsRaw=z(0), variance=z(0))
        self.expected = np.zeros((n, n), dtype=np.float32)
        for i in range(n):
            for j in range(n):
                dist = f32(math.hypot(i - n / 2.0, j - n / 2.0))
                with np.errstate(divide="ignore"):
                    inv = f32(1) / dist
                self.expected[i, j] = f32(np.arctan(inv, dtype=np.float32)) / self.vpad

    # grid_map_core getIndexFromPosition / checkIfPositionWithinMap
    def get_index(self, x, y):
        half = f64(0.5) * self.length
        ivx = ((f64(x) - half) - self.pos[0]) / self.res
        ivy = ((f64(y) - half) - self.pos[1]) / self.res
        row, col = trunc_int(-ivx), trunc_int(-ivy)
        ax = (f64(x) - self.pos[0]) - half
        ay = (f64(y) - self.pos[1]) - half
        tx = f64(-1.0) * ax + f64(0.0) * ay
        ty = f64(0.0) * ax + f64(-1.0) * ay
        inside = bool(tx >= 0.0 and ty >= 0.0 and tx < self.length and ty < self.length)
        return inside, row, col

    def reset(self):
        L = self.L
        for k in ("groundCandidates", "planeDist", "m2", "meanVariance", "pointsRaw", "points", "variance"):
            L[k][...] = 0
        L["minGroundHeight"][...] = FLT_MAX
        L["maxGroundHeight"][...] = FLT_MIN

    def insert(self, cloud, origin):
        L, n, cfg = self.L, self.n, self.cfg
        ox, oy, oz = f32(origin[0]), f32(origin[1]), f32(origin[2])
        cls = np.zeros(len(cloud), dtype=np.uint8)
        cell = np.full(len(cloud), -1, dtype=np.int32)
        with np.errstate(all="ignore"):
            for i in range(len(cloud)):
                x, y, z, ring = f32(cloud["x"][i]), f32(cloud["y"][i]), f32(cloud["z"][i]), int(cloud["ring"][i])
                dx, dy = f32(x - ox), f32(y - oy)
                sqdist = f32(f64(dx) * f64(dx) + f64(dy) * f64(dy))
                inside, r, c = self.get_index(x, y)
                if not inside or r < 0 or c < 0 or r >= n or c >= n:
                    continue
                cell[i] = r + c * n
                L["pointsRaw"][r, c] = f32(L["pointsRaw"][r, c] + f32(1))
                if ring > cfg["max_ring"] or sqdist < self.min_dist_sq:
                    cls[i] = 1
                    continue
                old = L["ground"][r, c]
                skip = False
                if f64(z) < f64(old) - f64(0.2):
                    vx, vy, vz = f32(x - ox), f32(y - oy), f32(z - oz)
                    ln = f32(np.sqrt(f32(f32(f32(vx * vx) + f32(vy * vy)) + f32(vz * vz))))
                    vx, vy, vz = f32(vx / ln), f32(vy / ln), f32(vz / ln)
                    len2 = f64(ln) * f64(ln)
                    step = 3
                    while True:
                        sx, sy, sz = f32(f32(step) * vx), f32(f32(step) * vy), f32(f32(step) * vz)
                        d2 = f64(sx) * f64(sx) + f64(sy) * f64(sy) + f64(sz) * f64(sz)
                        if not (d2 < len2 and vz < f32(-0.01)):
                            break
                        _, I0, I1 = self.get_index(f32(sx + ox), f32(sy + oy))
                        if not (I0 <= 0 or I1 <= 0 or I0 >= n - 1 or I1 >= n - 1):
                            r0, c0 = max(I0 - 1, 2), max(I1 - 1, 2)
                            blk = [L["groundpatch"][r0 + s % 3, c0 + s // 3] for s in range(9)]
                            if (f64(tree_sum(blk)) > cfg["min_outlier_detection_ground_confidence"]
                                    and L["groundpatch"][I0, I1] > f32(0.01)
                                    and f64(L["ground"][I0, I1]) >= f64(f32(sz + oz)) + f64(cfg["outlier_tolerance"])):
                                skip = True
                                break
                        step += 1
                if skip:
                    cls[i] = 2
                    continue
                cls[i] = 3
                pts = L["points"][r, c]
                pd = f32(z - oz)
                L["groundCandidates"][r, c] = f32(f64(f32(z + f32(pts * L["groundCandidates"][r, c]))) / (f64(pts) + f64(1.0)))
                if f64(L["meanVariance"][r, c]) == 0.0:
                    L["meanVariance"][r, c] = pd
                if not np.isnan(pd):
                    mean = L["meanVariance"][r, c]
                    delta = f32(pd - mean)
                    mean = f32(mean + f32(delta / f32(pts + f32(1))))
                    L["meanVariance"][r, c] = mean
                    L["planeDist"][r, c] = f32(f64(f32(pd + f32(pts * L["planeDist"][r, c]))) / (f64(pts) + f64(1.0)))
                    L["m2"][r, c] = f32(L["m2"][r, c] + f32(delta * f32(pd - mean)))
                L["maxGroundHeight"][r, c] = std_max(L["maxGroundHeight"][r, c], z)
                L["minGroundHeight"][r, c] = std_min(L["minGroundHeight"][r, c], f32(z - f32(0.0001)))
                L["points"][r, c] = f32(f64(pts) + f64(1.0))
        return cls, cell

    def _patch(self, S, i, j):
        L, n, cfg = self.L, self.n, self.cfg
        ci = S // 2
        blk = lambda name: [L[name][i - ci + s % S, j - ci + s // S] for s in range(S * S)]  # noqa: E731
        pts = blk("points")
        resf = f32(self.res)
        sqdist = f32((f64(i - n / 2.0) ** 2 + f64(j - n / 2.0) ** 2) * (f64(resf) * f64(resf)))
        expected = self.expected[i, j]
        psum = block_sum(pts)
        oldC, oldG = L["groundpatch"][i, j], L["ground"][i, j]
        gp = f64(cfg["ground_patch_detection_minimum_point_count_threshold"])
        if f64(psum) < std_max(np.floor(gp * f64(S) * f64(expected)), f64(3.0)):
            return
        df, mdf = f64(cfg["distance_factor"]), f64(cfg["minimum_distance_factor"])
        var_thr = f32(std_min(std_max(f64(sqdist) * (df * df), mdf * mdf), (mdf * f64(10)) * (mdf * f64(10))))
        var, mn = blk("variance"), blk("minGroundHeight")
        variance = var[ci + ci * S]
        localmin = min(mn)
        if pts[ci + ci * S] >= f32(cfg["point_count_cell_variance_threshold"]):
            max_var = variance
        else:
            max_var = f32(block_sum([f32(p * v) for p, v in zip(pts, var)]) / psum)
        groundlevel = f32(block_sum([f32(p * m) for p, m in zip(pts, mn)]) / psum)
        ground_diff = std_max(f32(f32(groundlevel - oldG) * f32(f32(2.0) * oldC)), f32(1.0))
        if f64(oldC) > 0.5 and f64(groundlevel) >= f64(oldG) + f64(cfg["outlier_tolerance"]):
            return
        if (f64(var_thr) > f64(max_var) * f64(max_var) and max_var > f32(0)
                and f64(psum) > f64(f32(f32(ground_diff * expected) * f32(S))) * gp):
            ocf = f64(cfg["occupied_cells_point_count_factor"])
            newC = f32(std_min(f64(psum) / ocf, f64(1.0)))
            L["ground"][i, j] = f32(f32(f32(groundlevel * newC) + f32(f32(oldC * oldG) * f32(2))) / f32(newC + f32(oldC * f32(2))))
            L["groundpatch"][i, j] = f32(std_min((f64(psum) / (ocf * f64(f32(2.0))) + f64(oldC)) / f64(2.0), f64(1.0)))
        elif localmin < oldG:
            L["ground"][i, j] = localmin
            L["groundpatch"][i, j] = std_min(f32(oldC + f32(0.1)), f32(0.5))

    def detect(self):
        L, n, cfg = self.L, self.n, self.cfg
        with np.errstate(all="ignore"):
            L["variance"][...] = (L["m2"] / (L["points"] + FLT_MIN)).astype(np.float32)
            resf = f32(self.res)
            pscd = f64(cfg["patch_size_change_distance"])
            for section in range(4):
                cs = 2 + section % 2 * (n // 2 - 2)
                rs = n // 2 if section >= 2 else 2
                ce = n // 2 + section % 2 * (n // 2 - 2)
                re_ = n - 2 if section >= 2 else n // 2
                for i in range(cs, ce):
                    for j in range(rs, re_):
                        sqdist = f32((f64(i - n / 2.0) ** 2 + f64(j - n / 2.0) ** 2) * (f64(resf) * f64(resf)))
                        self._patch(3 if f64(sqdist) <= pscd * pscd else 5, i, j)

    def _interp(self, x, y):
        L, cfg = self.L, self.cfg
        c = self.n // 2 - 1
        w = [L["groundpatch"][x - 1 + s % 3, y - 1 + s // 3] for s in range(9)]
        g = [L["ground"][x - 1 + s % 3, y - 1 + s // 3] for s in range(9)]
        height, occ = L["ground"][x, y], L["groundpatch"][x, y]
        wsum = f32(tree_sum(w) + FLT_MIN)
        avg = f32(tree_sum([f32(a * b) for a, b in zip(w, g)]) / wsum)
        L["ground"][x, y] = f32(f32(f32(f32(1.0) - occ) * avg) + f32(occ * height))
        fx, fy = f32(f32(x) - f32(c)), f32(f32(y) - f32(c))
        if (f64(fx) * f64(fx) + f64(fy) * f64(fy)) * (self.res * self.res) > f64(self.min_dist_sq):
            L["groundpatch"][x, y] = f32(std_max(f64(occ) - f64(occ) / f64(cfg["occupied_cells_decrease_factor"]), f64(0.001)))

    def spiral(self, base_z):
        L = self.L
        c = self.n // 2 - 1
        L["groundpatch"][c, c] = f32(1.0)
        L["ground"][c, c] = f32(f64(base_z))
        with np.errstate(all="ignore"):
            for i in range(c - 1, 0, -1):
                rp = i
                sl = (c - rp) * 2
                for side in range(2):
                    for pos in range(rp, rp + sl):
                        self._interp(pos if side % 2 else rp, rp if side % 2 else pos)
                rp += sl
                for side in range(2):
                    for pos in range(rp, rp - sl - 1, -1):
                        self._interp(pos if side % 2 else rp, rp if side % 2 else pos)

    def filter_cloud(self, cloud, origin=(0, 0, 0), base_z=0.0):
        L, n, cfg = self.L, self.n, self.cfg
        self.reset()
        cls, cell = self.insert(cloud, origin)
        self.detect()
        self.spiral(base_z)
        L["points"][...] = 0
        mdf = f64(cfg["minimum_distance_factor"]) * f64(5)
        th, oth = f64(cfg["miminum_point_height_threshold"]), f64(cfg["minimum_point_height_obstacle_threshold"])
        ox, oy = f32(origin[0]), f32(origin[1])
        label = np.zeros(len(cloud), dtype=np.uint8)
        index = np.full(len(cloud), -1, dtype=np.int32)
        k = 0
        with np.errstate(all="ignore"):
            for want in (3, 1):
                for i in np.nonzero(cls == want)[0]:
                    r, c = int(cell[i]) % n, int(cell[i]) // n
                    gh, var = f64(L["ground"][r, c]), L["variance"][r, c]
                    if n <= r + 3 or n <= c + 3:
                        continue
                    dxf, dyf = f32(f32(cloud["x"][i]) - ox), f32(f32(cloud["y"][i]) - oy)
                    dist = f32(np.sqrt(f64(dxf) * f64(dxf) + f64(dyf) * f64(dyf)))
                    tol = std_max(std_min((mdf * f64(dist)) / f64(var) * th, th), oth)
                    if tol + gh < f64(f32(cloud["z"][i])):
                        label[i] = 99
                        L["points"][r, c] = f32(L["points"][r, c] + f32(1))
                    else:
                        label[i] = 49
                    index[i] = k
                    k += 1
            for i in np.nonzero(cls == 2)[0]:
                label[i] = 49
                index[i] = k
                k += 1
        return dict(label=label, index=index, cls=cls, cell=cell, out_n=k)
