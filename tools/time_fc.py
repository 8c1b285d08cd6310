"""Time the small-K fc kernels against the library path at the shapes of the shipped configs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeprl_network_amd import ops


def timeit(f, n=20):
    for _ in range(3): f()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): f()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


for N, rows, F, tag in [(8, 4096, 15, 'cacc rollout x~'), (25, 1024, 60, 'grid rollout x~'), (25, 1024, 64, 'grid rollout msg'),
                        (8, 4096, 64, 'cacc rollout msg'), (8, 245760, 15, 'cacc update x~'), (25, 122880, 60, 'grid update x~')]:
    x = torch.randn(N, rows, F, device='cuda'); w = torch.randn(N, F, 64, device='cuda') * 0.1; b = torch.zeros(N, 64, device='cuda')
    y = torch.empty(N, rows, 64, device='cuda'); dy = torch.randn(N, rows, 64, device='cuda')
    t_k = timeit(lambda: ops.fc_fwd(x, w, b, 1, out=y), 20 if rows < 10000 else 3)
    t_g = timeit(lambda: ops.bias_act_(torch.bmm(x, w), b, 1, out=y), 20 if rows < 10000 else 3)
    line = '%-18s N=%d rows=%d F=%d: fwd kernel %.1f us, GEMM+bias_act %.1f us' % (tag, N, rows, F, t_k, t_g)
    if rows > 10000:
        t_b = timeit(lambda: ops.fc_bwd(x, y, dy, 1), 3)
        def lib():
            g = dy * (y > 0)
            return ops.wgrad(x, g), g.sum(1)
        line += '; bwd kernel %.1f us, mask + split wgrad + sum %.1f us' % (t_b, timeit(lib, 3))
    print(line)

# the heads' backward over the update batch (thin linear layer, 64 -> 5 columns)
N, rows, O = 8, 245760, 5
h = torch.randn(N, rows, 64, device='cuda'); dy = torch.randn(N, rows, O, device='cuda'); w = torch.randn(N, 64, O, device='cuda')
t_t = timeit(lambda: ops.thin_linear_bwd(h, dy, w), 3)
print('thin_linear_bwd N=%d rows=%d O=%d: %.1f us (h read + dh written: %.0f MB -> %.2f TB/s)' % (N, rows, O, t_t, 2 * h.numel() * 4 / 1e6,
                                                                                               2 * h.numel() * 4 / t_t / 1e6))
