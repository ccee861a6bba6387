"""The reference call shape -- one cloud per call -- measured from the host: the synchronous call, the fused call with all layers
(plain and registered planes), the wire-to-wire call, the device-resident single-cloud batch with and without graph replay.
   python tools/host_call_probe.py   -> JSON (ms per call, best of 3 passes)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import api, synth


def best(fn, seq, passes=3):
    t = float("inf")
    for _ in range(passes):
        t0 = time.perf_counter()
        for c in seq:
            fn(c)
        t = min(t, (time.perf_counter() - t0) / len(seq))
    return round(t * 1e3, 4)


def main():
    clouds = [synth.hdl64_cloud(seed=20240113 + k) for k in range(4)]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    org = (0.0, 0.0, 0.0)
    res = {"lib": os.environ.get("GROUNDGRID_HIP_LIB", "default"), "graph_env": os.environ.get("GG_GRAPH", "unset"), "points": [len(c) for c in clouds]}
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=1, max_points=stride)
    seq = [clouds[k % 4] for k in range(48)]
    for c in seq[:6]:
        seg.filter_cloud(c, org, -1.73)
    res["sync_ms"] = best(lambda c: seg.filter_cloud(c, org, -1.73, reuse_buffers=True), seq)
    plain, pinned = seg.alloc_layers(register=False), seg.alloc_layers(register=True)
    state = {k: plain[k] for k in ("ground", "groundpatch", "points", "pointsRaw")}
    res["fused_all_layers_ms"] = best(lambda c: seg.filter_cloud_with_layers(c, org, -1.73, plain, reuse_buffers=True), seq[:32])
    res["fused_all_layers_registered_ms"] = best(lambda c: seg.filter_cloud_with_layers(c, org, -1.73, pinned, reuse_buffers=True), seq[:32])
    res["fused_state_layers_ms"] = best(lambda c: seg.filter_cloud_with_layers(c, org, -1.73, state, reuse_buffers=True), seq[:32])

    def two_calls(c):
        seg.filter_cloud(c, org, -1.73, reuse_buffers=True)
        seg.map(0).layers()

    res["two_calls_all_layers_ms"] = best(two_calls, seq[:32])
    wires = {id(c): api.to_pc2(c).tobytes() for c in clouds}
    res["pc2_in_pc2_out_ms"] = best(lambda c: seg.filter_cloud_pc2_out(wires[id(c)], len(c), 18, (0, 4, 8, 16), org, -1.73), seq[:32])
    res["pc2_in_labels_out_ms"] = best(lambda c: seg.filter_cloud_pc2(wires[id(c)], len(c), 18, (0, 4, 8, 16), org, -1.73), seq[:32])
    res["graph_replays_host_calls"] = seg.debug_set_tuning("graph_replays", 0)

    # device-resident input, one cloud per launch
    host = np.zeros((1, stride), dtype=api.POINT16_DTYPE)
    host[0, : len(clouds[0])] = api.pack16(clouds[0])
    pts = torch.from_numpy(host.view(np.uint8).reshape(1, stride, 16)).cuda()
    n, o3, bz = [len(clouds[0])], np.zeros((1, 3), np.float32), np.full(1, -1.73)
    side = torch.cuda.Stream()
    for graphs in (1, 0):
        seg.debug_set_tuning("graphs", graphs)
        with torch.cuda.stream(side):
            out = None
            for _ in range(6):
                out = seg.filter_batch(pts, n, o3, bz, out=out)
            seg.synchronize()
            t = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(40):
                    out = seg.filter_batch(pts, n, o3, bz, out=out)
                seg.synchronize()
                t = min(t, (time.perf_counter() - t0) / 40)
        res["device_resident_one_cloud_ms_graph" if graphs else "device_resident_one_cloud_ms_eager"] = round(t * 1e3, 4)
    seg.debug_set_tuning("graphs", 0)
    seg.set_flags(profile=True)
    with torch.cuda.stream(side):
        out = None
        for _ in range(4):
            out = seg.filter_batch(pts, n, o3, bz, out=out)
        seg.synchronize()
        seg.kernel_times(reset=True)
        for _ in range(20):
            out = seg.filter_batch(pts, n, o3, bz, out=out)
        seg.synchronize()
    res["kernel_ms"] = {k: round(v[0] / max(1, v[1]), 4) for k, v in seg.kernel_times().items()}
    res["kernel_sum_ms"] = round(sum(res["kernel_ms"].values()), 4)
    seg.release_layers(pinned)
    seg.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
