"""Micro-benchmark: wgrad GEMM s^T dz (K = T*E rows) direct vs manual split-K batches."""
import torch


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


N, rows = 8, 245760
for M in (128, 64):
    s = torch.randn(N, rows, M, device='cuda')
    dz = torch.randn(N, rows, 256, device='cuda')
    ref = torch.bmm(s.transpose(1, 2), dz)
    print('M=%d direct: %.0f us' % (M, timeit(lambda: torch.bmm(s.transpose(1, 2), dz))))
    for S in (4, 8, 16, 32, 64):
        sv = s.view(N * S, rows // S, M)
        dv = dz.view(N * S, rows // S, 256)
        f = lambda: torch.bmm(sv.transpose(1, 2), dv).view(N, S, M, 256).sum(1)
        err = (f() - ref).abs().max().item()
        print('  split %2d: %.0f us (max err %.2e)' % (S, timeit(f), err))
    # fwd-like orientation for comparison
    w = torch.randn(N, M, 256, device='cuda')
    print('  fwd s@w: %.0f us;  dgrad dz@w^T: %.0f us' % (timeit(lambda: torch.bmm(s, w)), timeit(lambda: torch.bmm(dz, w.transpose(1, 2)))))
