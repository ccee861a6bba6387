"""Practical HBM ceilings of the box (torch kernels): fill (write only), copy (read + write), sum (read only)."""
import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); ts = t(lambda: a.sum())
    print(f"{mb:5d} MB: fill {n*4/tf/1e9:7.0f} GB/s   copy {2*n*4/tc/1e9:7.0f} GB/s (r+w)   sum {n*4/ts/1e9:7.0f} GB/s")
