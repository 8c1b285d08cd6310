"""Time the update's fused heads + loss pass (nmarl_heads_loss) against the chain it replaces, on a BASELINE shape.
    python tools/time_heads_loss.py [N E T A]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeprl_network_amd import ops

N, E, T, A = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (8, 4096, 60, 4)
rows, H = T * E, 64
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()            # noqa: E731
h = torch.tanh(r(N, rows, H))
m = 2 if N == 8 else 4
nbr = torch.tensor([[(i + k + 1) % N for k in range(m)] for i in range(N)], dtype=torch.int32).cuda()
pi_w, pi_b, v_w, v_b = r(N, H, A) * .1, r(N, A) * .1, r(N, H + m * A, 1) * .1, r(N, 1) * .1
action = torch.randint(0, A, (rows, N), generator=g).to(torch.uint8).cuda()
adv, R = r(N, rows), r(N, rows)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def chain():
    hh = h.detach().requires_grad_(True)
    ps = [p.detach().requires_grad_(True) for p in (pi_w, pi_b, v_w, v_b)]
    logits, v = ops.heads(hh, *ps, action, nbr, A)
    per, _ = ops.a2c_loss(logits, v, action, adv, R, 0.5, 0.01)
    per.sum().backward()


for want_dh in (True, False):
    print('fused, dh %s: %.1f us' % ('written' if want_dh else 'not written',
                                     timed(lambda: ops.heads_loss(h, pi_w, pi_b, v_w, v_b, action, nbr, A, adv, R, 0.5, 0.01, want_dh=want_dh))))
print('chain (GEMM + loss fwd/bwd + thin_bwd): %.1f us' % timed(chain))
