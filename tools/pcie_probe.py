"""How long do the host call's transfers take on this box?  2 MB up (one copy, two copies on two streams, four on four), 0.63 MB down.
   python tools/pcie_probe.py  -> JSON (microseconds, best of 200, host clock around issue + synchronize)"""
import json, time, torch

def best(fn, n=200):
    t = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t = min(t, time.perf_counter() - t0)
    return round(t * 1e6, 1)

res = {}
for mb, name in ((2.0, "up_2MB"), (1.0, "up_1MB"), (0.63, "down_0.63MB"), (0.5, "up_0.5MB")):
    n = int(mb * (1 << 20))
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(4)]
    if name.startswith("up"):
        res[name] = best(lambda: d.copy_(h, non_blocking=True))
        for k in (2, 4):
            def split():
                c = n // k
                for i in range(k):
                    with torch.cuda.stream(streams[i]):
                        d[i * c:(i + 1) * c].copy_(h[i * c:(i + 1) * c], non_blocking=True)
            res[f"{name}_in_{k}_streams"] = best(split)
    else:
        res[name] = best(lambda: h.copy_(d, non_blocking=True))
        def split2():
            c = n // 2
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    h[i * c:(i + 1) * c].copy_(d[i * c:(i + 1) * c], non_blocking=True)
        res[name + "_in_2_streams"] = best(split2)
res["empty_sync"] = best(lambda: None)
print(json.dumps(res))
