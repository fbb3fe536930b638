"""HBM bandwidth ceilings as seen by simple torch kernels (fill = write only, copy = read + write, sum = read only)."""
import torch
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for mb in (256, 740, 1480, 4000):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    tf = timeit(lambda: x.fill_(1.0)); tc = timeit(lambda: y.copy_(x)); ts = timeit(lambda: x.sum())
    print(f"{mb:5d} MB: fill {mb/tf/1e3:6.2f} TB/s   copy (r+w) {2*mb/tc/1e3:6.2f} TB/s   sum {mb/ts/1e3:6.2f} TB/s")
