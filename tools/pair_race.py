"""Hunt an intermittent mismatch of the pair sweep: one cloud, frame by frame against the oracle; prints the first wrong cells in sweep order."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth
from oracle import oracle

cloud = synth.hdl64_cloud(seed=20240113)
stride = (len(cloud) + 63) // 64 * 64
host = np.zeros((1, stride), dtype=api.POINT16_DTYPE); host[0, :len(cloud)] = api.pack16(cloud)
pts = torch.from_numpy(host.view(np.uint8).reshape(1, stride, 16)).cuda()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for attempt in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=stride)
    ref = oracle.OracleMap(120.0, 0.33)
    out = None
    for f in range(frames):
        out = seg.filter_batch(pts, [len(cloud)], np.zeros((1, 3), np.float32), np.full(1, -1.73), out=out)
        seg.synchronize()
        ref.filter_cloud(cloud, (0.0, 0.0, 0.0), -1.73)
        g, G = seg.map(0)["ground"], ref.layer("ground")
        if not np.array_equal(g, G, equal_nan=True):
            n = g.shape[0]; c = n // 2 - 1
            bad = np.argwhere(~((g == G) | (np.isnan(g) & np.isnan(G))))
            rows = []
            for x, y in bad.tolist():
                dx, dy = x - c, y - c; r = max(abs(dx), abs(dy))
                if dx == -r and dy < r: side, k = 'A', y - (c - r)
                elif dx == r: side, k = 'C', (c + r) - y
                elif dy == -r: side, k = 'B', x - (c - r)
                else: side, k = 'D', (c + r) - x
                rows.append((r, side, k, float(g[x, y]), float(G[x, y])))
            rows.sort()
            print(f"attempt {attempt} frame {f}: {len(bad)} wrong cells; first: {rows[:12]}", flush=True)
            break
    else:
        print(f"attempt {attempt}: {frames} frames ok", flush=True)
    seg.close()
