"""Practical streaming bandwidth of the box (context for the roofline fractions): torch sum / copy_ / fill_ over fp64 arrays."""
import torch, time
for n in (150_000_000, 1_600_000_000):
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for name, f, bytes_ in (("sum (read)", lambda: x.sum(), 8*n), ("copy (read+write)", lambda: y.copy_(x), 16*n), ("fill (write)", lambda: y.fill_(1.0), 8*n)):
        f(); torch.cuda.synchronize()
        t=time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize()
        dt=(time.perf_counter()-t)/5
        print(f"n={n:>11} {name:18s} {bytes_/dt/1e12:6.2f} TB/s ({dt*1e3:.3f} ms)")
    del x, y
