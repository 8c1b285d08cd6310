"""Practical streaming ceilings of this MI355X for the access mixes the env kernels have:
fill (write only), copy (1R:1W), read-reduce (read only).  HIP events, 1 GiB fp32 buffers."""
import torch
n = 1 << 28
x = torch.ones(n, device='cuda'); y = torch.empty(n, device='cuda')
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
us = timeit(lambda: y.fill_(1.5)); print('fill   1 GiB: %.1f us  %.2f TB/s' % (us, n * 4 / us / 1e6))
us = timeit(lambda: y.copy_(x)); print('copy   1 GiB: %.1f us  %.2f TB/s (R+W)' % (us, 2 * n * 4 / us / 1e6))
us = timeit(lambda: x.sum()); print('sum    1 GiB: %.1f us  %.2f TB/s' % (us, n * 4 / us / 1e6))
us = timeit(lambda: torch.add(x, 1.0, out=y)); print('add    1 GiB: %.1f us  %.2f TB/s (R+W)' % (us, 2 * n * 4 / us / 1e6))
