"""Timing of the coupled nets' two hand-off kernels at the BASELINE shapes (HIP events around repeated launches):
  nmarl_lstm_bptt_coupled   NeurComm 8 x 4096, T = 60 (configs[2] / [4])  and  CommNet grid 25 x 1024, T = 120 (configs[3])
  nmarl_lstm_step_x_msg     head kind 3 (one launch per lock-step), same two shapes, as a 60-launch hipGraph
python tools/time_coupled.py [bptt] [step]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
from deeprl_network_amd import ops  # noqa: E402
from test_gpu_ops import _forward_cells, _topology  # noqa: E402

H = 64
what = [a for a in sys.argv[1:] if not a.startswith('-')] or ['bptt', 'step']
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()            # noqa: E731
rd = lambda *s: torch.randn(*s, device='cuda')                # noqa: E731


def events(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def bptt(kind, topo, N, E, T):
    nbr_idx, _ = ops.neighbor_table(_topology(N, topo), 'cuda')
    m_max = nbr_idx.shape[1]
    K = H * m_max if kind == ops.COUPLED_NC else H
    G = torch.cat([torch.sigmoid(rd(N, T, E, 3 * H)), torch.tanh(rd(N, T, E, H))], dim=-1)
    C, D = rd(N, T + 1, E, H) * 0.5, rd(N, T, E, H)
    dZ, D1 = torch.empty(N, T, E, 4 * H, device='cuda'), torch.empty(N, T, E, H, device='cuda')
    done = torch.zeros(T, E, device='cuda')
    wxm, wh, wmsg = rd(N, H, 4 * H) * 0.1, rd(N, H, 4 * H) * 0.1, rd(N, K, H) * 0.15
    S = torch.relu(rd(N, T, E, 3 * H))
    ws, wm = (wxm, wh, ops.lstm_bptt_wimage(wxm, wh)), (wmsg, ops.lstm_bptt_msg_wimage(wmsg))
    rev = ops.reverse_neighbor_table(nbr_idx, kind)
    mask = S[..., 2 * H:] if kind == ops.COUPLED_NC else None
    us = events(lambda: ops.bptt_coupled(kind, rev, m_max, G, C, done, D, ws, wm, mask, dZ, D1), 5)
    ops.check_coupled_status()
    src = float((nbr_idx >= 0).sum().item()) / N
    row = 1024 + 256 + 256 + (256 if mask is not None else 0) + src * 256 + 1024 + 256 + K * 4
    print('bptt_coupled kind %d %s N %d E %d T %d: %.1f us = %.1f us/step, %.0f B/row algorithmic -> %.2f TB/s = %.3f of 8 TB/s'
          % (kind, topo, N, E, T, us, us / T, row, N * T * E * row / us / 1e6, N * T * E * row / us / 1e6 / 8.0))


def step(kind, topo, N, E, A, with_ob):
    nbr, _ = ops.neighbor_table(_topology(N, topo), 'cuda')
    m_max = nbr.shape[1]
    KXg = 2 * H if kind == 1 else 0
    KX, Km = KXg + H, (H * m_max if kind == 1 else H)
    h, c = r(N, E, H) * 0.3, r(N, E, H) * 0.3
    wx, wh, b = r(N, KX, 4 * H) * 0.15, r(N, H, 4 * H) * 0.2, r(N, 4 * H) * 0.1
    w_msg, b_msg = r(N, Km, H) * 0.15, r(N, H) * 0.1
    img, mimg = ops.lstm_wimage(wx, wh), ops.lstm_msg_wimage(w_msg)
    pi_w, pi_b, v_w, v_b = r(N, H, A), r(N, A), r(N, H + m_max * A, 1), r(N, 1)
    done = torch.zeros(E, device='cuda')
    ho, co, gates = torch.empty_like(h), torch.empty_like(c), torch.empty(N, E, 4 * H, device='cuda')
    pi, act, v = torch.empty(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda'), torch.empty(N, E, device='cuda')
    slot = torch.relu(r(N, E, KX))
    sync = ops.step_sync_words(N, E, 'cuda')
    msg = dict(kind=kind, nbr_idx=nbr, w_msg=w_msg, b_msg=b_msg, img=mimg, sync=sync)
    flops = N * E * (2 * (KX + H) * 4 * H + 2 * (2 * H) * 4 * H + 2 * 2 * Km * H)
    if kind == 1:
        msg['out'] = slot[:, :, KXg:]
    else:
        msg['enc'], msg['out'] = torch.zeros(N, E, H, device='cuda'), torch.zeros(N, E, H, device='cuda')
        if with_ob:
            Fo = 12
            nbr_self = torch.cat([torch.arange(N, dtype=torch.int32, device='cuda').view(-1, 1), nbr], dim=1)
            w_ob, b_ob = r(N, Fo * nbr_self.shape[1], H) * 0.3, r(N, H) * 0.1
            msg['ob'] = dict(x=r(E, N, Fo), nbr=nbr_self, img=ops.lstm_ob_wimage(w_ob, torch.zeros(N, 64, H, device='cuda')), b=b_ob)
            flops += N * E * 2 * H * H

    def one():
        ops.lstm_step_policy_value(h, None, b, None, None, c, done, pi_w, pi_b, pi, act, v_w, v_b, nbr, A, v, mode=2,
                                   xs=(slot[:, :, :KXg] if KXg else None, None, img, None, msg), h_out=ho, c_out=co, gates=gates,
                                   defer_action_term=True)
    n = 60
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        one()
    torch.cuda.current_stream().wait_stream(s_)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            one()
    us = events(gr.replay, 10) / n
    ops.check_coupled_status()
    print('lstm_step_x<4,%d> %s N %d E %d%s: %.1f us per launch (60-launch graph) = %.1f TFLOP/s = %.3f of 157.3'
          % (kind, topo, N, E, ' +encoder' if with_ob else '', us, flops / us / 1e6, flops / us / 1e6 / 157.3))


if 'bptt' in what:
    bptt(ops.COUPLED_NC, 'line', 8, 4096, 60)
    bptt(ops.COUPLED_IC3, 'grid', 25, 1024, 120)
    bptt(ops.COUPLED_IC3, 'line', 8, 4096, 60)
if 'step' in what:
    step(1, 'line', 8, 4096, 4, False)
    step(2, 'grid', 25, 1024, 5, True)
    step(2, 'line', 8, 4096, 4, False)
