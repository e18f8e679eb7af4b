"""Plain streams of the BASELINE step's byte volumes (context for its kernels' times): torch copy_ of 2 x 96 MB (the warp kernel's counter traffic is 192 MB per launch),
torch sum over 94 MB (what the Gram kernel reads), timed by HIP events around 200 back-to-back launches."""
import torch
def timed(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for mb, what in ((96, "copy_ (read + write, 2 x 96 MB)"), (94, "sum (read 94 MB)"), (122, "copy_ (2 x 122 MB = the 244-B model at 1 M events)")):
    n = mb * 1_000_000 // 8
    x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    f = (lambda: y.copy_(x)) if "copy" in what else (lambda: x.sum())
    us = timed(f)
    moved = (2 if "copy" in what else 1) * n * 8
    print(f"{what:55s} {us:7.1f} us per launch  ({moved / us / 1e6:5.2f} TB/s)")
us = timed(lambda: torch.empty(1, device="cuda").fill_(0.0))
print(f"{'one-element fill_ (launch floor, back to back)':55s} {us:7.1f} us per launch")
