"""Time the fused LSTM step kernels at the bench shape (N = 8, E = 4096, H = 64):
  * recurrent-only step (nmarl_lstm_step_fused: addend from a separate s @ Wx library GEMM, timed beside it),
  * the x-side step (nmarl_lstm_step_x: K = KX + 64 inside), plain / with gates / with the policy+value heads (kind 3).
    python tools/time_fused.py [E] [KX]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deeprl_network_amd import ops  # noqa: E402

N, H, A = 8, 64, 4
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
KX = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()          # noqa: E731
h, c, z1 = r(N, E, H), r(N, E, H), r(N, E, 4 * H)
x, wx, wh = r(N, E, KX) * 0.5, r(N, KX, 4 * H) * 0.15, r(N, H, 4 * H) * 0.2
b, done = torch.zeros(N, 4 * H).cuda(), torch.zeros(E).cuda()
pi_w, pi_b, v_w, v_b = r(N, H, A), r(N, A), r(N, H + 2 * A, 1), r(N, 1)
nbr = torch.tensor([[max(i - 1, 0), min(i + 1, N - 1)] for i in range(N)], dtype=torch.int32).cuda()
co, ho = torch.empty_like(c), torch.empty_like(h)
gates = torch.empty(N, E, 4 * H, device='cuda')
pi, act, v = torch.empty(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda'), torch.empty(N, E, device='cuda')
img = ops.lstm_wimage(wx, wh)
zbuf = torch.empty(N, E, 4 * H, device='cuda')


def timed(name, f, n=20, reps=10):
    for _ in range(3):
        f()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            f()
    gr.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        gr.replay()
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / (n * reps)
    print('%-58s %7.2f us' % (name, us))
    return us


flops_x = 2.0 * N * E * (KX + H) * 4 * H
print('N=%d E=%d KX=%d: x-side step = %.2f GFLOP -> %.1f us at the 157.3 TFLOP/s fp32 matrix peak' % (N, E, KX, flops_x / 1e9, flops_x / 157.3e6))
timed('library GEMM s @ Wx', lambda: torch.bmm(x, wx, out=zbuf))
timed('recurrent-only step (addend given)', lambda: ops.lstm_step_fused(h, wh, b, z1, None, c, done, None, co, ho))
timed('recurrent-only step + gates', lambda: ops.lstm_step_fused(h, wh, b, z1, None, c, done, gates, co, ho))
timed('recurrent-only policy+value (kind 3)', lambda: ops.lstm_step_policy_value(h, wh, b, z1, None, c, done, pi_w, pi_b, pi, act, v_w, v_b, nbr, A, v, mode=2))
u = timed('x-side step', lambda: ops.lstm_step_fused(h, None, b, None, None, c, done, None, co, ho, xs=(x, None, img)))
print('    -> %.1f TFLOP/s = %.2f of the fp32 matrix peak' % (flops_x / u / 1e6, flops_x / u / 157.3e6))
timed('x-side step + gates', lambda: ops.lstm_step_fused(h, None, b, None, None, c, done, gates, co, ho, xs=(x, None, img)))
u3 = timed('x-side policy+value (kind 3)', lambda: ops.lstm_step_policy_value(h, None, b, None, None, c, done, pi_w, pi_b, pi, act, v_w, v_b, nbr, A, v, mode=2, xs=(x, None, img)))
f3 = flops_x + 2.0 * N * E * H * 4 * H
print('    -> %.1f TFLOP/s = %.2f of the fp32 matrix peak (incl. the value re-step)' % (f3 / u3 / 1e6, f3 / u3 / 157.3e6))
timed('weight image rebuild', lambda: ops.lstm_wimage(wx, wh, out=img))

# ---- the reverse step: cell_bwd + dgrad GEMM vs the fused BPTT step
G = torch.cat([torch.sigmoid(r(N, E, 3 * H)), torch.tanh(r(N, E, H))], dim=-1)
cp, cn, dh, dh2, dc = r(N, E, H), r(N, E, H), r(N, E, H), r(N, E, H), r(N, E, H)
dz, dcp, dhd, dx = torch.empty(N, E, 4 * H, device='cuda'), torch.empty(N, E, H, device='cuda'), torch.empty(N, E, H, device='cuda'), torch.empty(N, E, H, device='cuda')
wxm = r(N, H, 4 * H) * 0.2
wh_t = wh.transpose(1, 2)
timed('cell_bwd', lambda: ops.cell_bwd(G, cp, cn, done, dh, dc, dz, dcp, dh2=dh2))
timed('dgrad GEMM dz @ wh^T', lambda: torch.bmm(dz, wh_t, out=dhd))
ws0 = (None, wh, ops.lstm_bptt_wimage(None, wh))
ws1 = (wxm, wh, ops.lstm_bptt_wimage(wxm, wh))
ub = timed('fused BPTT step (dh only)', lambda: ops.bptt_step(G, cp, cn, done, dh, dh2, dc, ws0, dz, dcp, dhd, True))
nb = N * E * (4 * H + 5 * H + 4 * H + 2 * H) * 4
print('    -> %.1f MB algorithmic -> %.2f TB/s' % (nb / 1e6, nb / ub / 1e6))
timed('fused BPTT step ([dx | dh], relu mask)', lambda: ops.bptt_step(G, cp, cn, done, dh, dh2, dc, ws1, dz, dcp, dhd, True, dx=dx, mask=cp))

# ---- the whole reverse recurrence in one launch
T = 60
Gs = torch.cat([torch.sigmoid(r(N, T, E, 3 * H)), torch.tanh(r(N, T, E, H))], dim=-1)
Cs, Ds, dZs = r(N, T + 1, E, H), r(N, T, E, H), torch.empty(N, T, E, 4 * H, device='cuda')
dones = torch.zeros(T, E, device='cuda')
us = timed('BPTT sequence, T = 60, one launch', lambda: ops.bptt_seq(Gs, Cs, dones, Ds, ws0[2], dZs), reps=5)
nbs = N * E * (4 * H + H + H + 4 * H) * 4 * T
print('    -> %.1f us per step; %.1f MB algorithmic per step -> %.2f TB/s' % (us / T, nbs / T / 1e6, nbs / us / 1e6))
