"""How long does a plain device copy with the fused LSTM step's traffic (65.5 MB in+out) take inside a hipGraph?"""
import torch
n = 8 * 4096 * 256                                     # floats: 33.5 MB read + 33.5 MB written = the step's 67 MB
x, y = torch.randn(n, device='cuda'), torch.empty(n, device='cuda')
f = lambda: y.copy_(x)
for _ in range(5): f()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): f()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): f()
g.replay(); torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10): g.replay()
t1.record(); torch.cuda.synchronize()
us = t0.elapsed_time(t1) * 1e3 / 200
print('copy of %.1f MB each way: %.2f us per launch = %.2f TB/s' % (n * 4 / 1e6, us, 2 * n * 4 / us / 1e6))
