"""bench.py -- env-steps/s of the batched rollout + A2C update on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
     or plainly -- without a torchrun environment the script re-launches itself with N ranks)

Workload (BASELINE.json configs[1]): CACC catch-up, 8 agents x 4096 lock-stepped replicas per GPU,
IA2C-FP (config/config_ia2c_fp_catchup.ini), fp32, synthetic (Philox initial conditions, random-init
sqrt(2)-orthogonal weights, actions sampled from the live policy).

A "step" = one n_step batch of the hot path over all replicas: 60 lock-steps of
{policy step, action draw, value re-step (reference quirk Q1), CACC env kernel, transition store},
the bootstrap value, the n-step return scan and ONE A2C update (unroll + loss + backward +
[RCCL all-reduce] + clip/RMSProp).  metric = agents x replicas x lock-steps / second, whole job.

Extra objects on the JSON line (tier contract):
  roofline      the DOMINANT kernel of the batch, the fused MFMA LSTM lock-step: algorithmic flops per launch / average
                launch duration measured live with HIP events, against the dense fp32 matrix peak; see `roofline.how`.
  roofline_env_step[_large_E]   the CACC step kernel (north_star's HBM roofline): algorithmic bytes per launch (B_alg,
                DESIGN.md) / average launch duration, at the workload size and at E = 2^21.
  cpu_baseline  the reference-equivalent E=1 CPU loop (oracle/trainer_ref.py, kind "port") timed
                on one host core on a bounded sample (rank 0, N=1 only).
"""
import argparse
import configparser
import json
import os
import sys
import time

# GEMM auto-tuning of the library products (PyTorch TunableOp over hipBLASLt / rocBLAS): each distinct GEMM shape is
# tuned once, at its first (untimed, warm-up) occurrence; the choices go to a per-device scratch file.  Must be set before
# torch initialises its BLAS handles.  `NMARL_BENCH_TUNABLEOP=0` (or --no-tunableop) measures the untuned library.
if os.environ.get('NMARL_BENCH_TUNABLEOP', '1') != '0' and '--no-tunableop' not in sys.argv:
    os.environ.setdefault('PYTORCH_TUNABLEOP_ENABLED', '1')
    os.environ.setdefault('PYTORCH_TUNABLEOP_FILENAME', '/tmp/nmarl_tunableop_%d.csv')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_AGENT = 8


def b_alg(obs_floats):
    """Algorithmic HBM bytes of one replica-step of the CACC kernel:
    per vehicle: read h,v (8) + action (1); write h,v,u (12) + observation; per replica: read
    t, v0_init, collided (9), write t, collided, done, global_reward and the scalar reward (14)."""
    return N_AGENT * (8 + 1 + 12) + obs_floats * 4 + 9 + 14


# compact observation (5 floats / vehicle) = the survey's 347 B (+4: the [E] reward vector);
# the product writes the 'ia2c' pre-gathered observation, sum n_s = 110 floats = 440 B
# (declared variant of SURVEY.md 8d; the 2 zero pad slots of the edge vehicles are NOT counted).
B_ALG_GATHERED = b_alg(110)
B_ALG_COMPACT = b_alg(40)       # 351 B: what the batched engine's env kernel moves (p.compact_obs)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default=os.path.join(ROOT, 'config', 'config_ia2c_fp_catchup.ini'))
    ap.add_argument('--envs', type=int, default=0, help='replicas per GPU (default: num_envs of the ini)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-port-worker', type=int, default=0, help='internal: run N batches of the CPU port, print the timing')
    ap.add_argument('--tune-only', action='store_true', help='internal: run one batch to tune the library GEMMs, print nothing')
    ap.add_argument('--no-tunableop', action='store_true', help='do not auto-tune the library GEMMs (PyTorch TunableOp)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the other BASELINE configs (NeurComm slow-down / catch-up, CommNet grid) measured after the headline')
    ap.add_argument('--other-steps', type=int, default=10, help='timed n_step batches per other config')
    ap.add_argument('--cpu-batches', type=int, default=100,
                    help='n_step batches of the E=1 CPU baseline (100 = 6000 env steps of BASELINE configs[0], ~10 s on one core)')
    return ap.parse_args()


def measure_step_kernel(env, actions_tape, reps=20, encode=None):
    """Average duration of one nmarl_cacc_step launch: a hipGraph of len(tape) back-to-back
    launches (real rollout state, the batch's own action tape, auto-reset on) bracketed by two
    HIP events on the launch stream; includes the ~1.5 us graph-node gaps, i.e. an upper bound."""
    tensors = env.state_tensors()
    state = [t.clone() for t in tensors]
    n = actions_tape.shape[0]

    def body():
        for k in range(n):
            if encode is None:
                env.step(actions_tape[k], auto_reset=True)
            else:
                env.step(actions_tape[k], auto_reset=True, encode=encode)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()                                  # untimed: the first replay also uploads the graph
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * n)
    for t, sv in zip(tensors, state):
        t.copy_(sv)
    return us


MFMA_F32_PEAK_TFLOPS = 157.3     # dense fp32 matrix peak (MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, no xf32 on gfx950)


def measure_lstm_step(model, n=60, reps=10):
    """Average duration of the DOMINANT kernel of the batch, the fused MFMA LSTM lock-step of the rollout, on the model's
    own shapes and weights: hipGraph of n launches between HIP events on the launch stream.
    x-side nets (IA2C / IA2C-FP / ConseNet): nmarl_lstm_step_x with the policy + value heads (kind 3): per (agent,
    replica) row 2*(KX+64)*256 flops of the policy step + 2*64*256 of the value re-step (quirk Q1), all on
    v_mfma_f32_16x16x4_f32.  Other nets: the recurrent-only step (2*64*256 flops per row).
    Returns (us per launch, flops per launch, algorithmic HBM bytes per launch, kernel name)."""
    from deeprl_network_amd import ops
    p = model.policy
    N, E, H = model.h_fw.shape
    A = model.n_a
    dev = model.device
    h, c = torch.randn(N, E, H, device=dev) * 0.3, torch.randn(N, E, H, device=dev) * 0.3
    done = torch.zeros(E, device=dev)
    wh, b = p.params[p.k_wh], p.params[p.k_b]
    if p.can_save_acts:
        p.refresh_wimage()
    if p.can_save_acts and p.fused_pv:
        KX = p.params[p.k_wx].shape[1]
        x = torch.relu(torch.randn(N, E, KX, device=dev))
        pi, act, v = torch.empty(N, E, A, device=dev), torch.zeros(E, N, dtype=torch.uint8, device=dev), torch.empty(N, E, device=dev)
        gates = torch.empty(N, E, 4 * H, device=dev)
        ho, co = torch.empty_like(h), torch.empty_like(c)

        in_k = bool(getattr(model, 'save_acts', False)) and p.enc_in_kernel(E, getattr(model, 'compact_obs', False))
        kw = dict(ob=dict(x=model.buf_x[0], fp=model.fp)) if in_k else {}

        def body():
            for _ in range(n):
                p.step_policy_value(x, h, c, done, pi, act, v, h_out=ho, c_out=co, gates=gates, defer_action_term=True,
                                    mode=ops.SAMPLE_PHILOX, seed=1, env_id_base=0, step=0, **kw)
        # (the in-kernel input encoders' 2 * 23 * 64 flops per row are not counted: the figure stays comparable across rounds)
        flops = N * E * (2 * (KX + H) * 4 * H + 2 * H * 4 * H)
        # read x (in-kernel encoders: the 23 encoder inputs instead), h, c; write h', c', gates, pi, v, action (+ the encoded input)
        nbytes = N * E * ((((23 if KX > H else 15) + KX) if in_k else KX) * 4 + 2 * H * 4 + 2 * H * 4 + 4 * H * 4 + A * 4 + 4 + 1)
        two = in_k and p._enc_spec(model.buf_x[0], model.fp, None).get('w_fp') is not None      # (one encoder: IA2C / ConseNet, ENC 2)
        name = 'lstm_step_x_kernel<3,0,%d> (nmarl_lstm_step_x%s, policy + value heads%s)' % (
            (1 if two else 2) if in_k else 0, '_enc' if in_k else '', ', input encoder%s inside' % ('s' if two else '') if in_k else '')
    elif p.can_save_acts and p.pv_one_launch(E):
        # coupled nets, one launch per lock-step: policy step (message term in the pre-phase), in-launch hand-off of the new h,
        # value re-step from the kept x-side part + the re-computed message columns + the new h (head kind 3 + message term)
        KX = p.params[p.k_wx].shape[1]
        Km = p.params['w_msg'].shape[1]
        enc = p.encode(model.buf_x[0], model.fp)
        pi, act, v = torch.empty(N, E, A, device=dev), torch.zeros(E, N, dtype=torch.uint8, device=dev), torch.empty(N, E, device=dev)
        gates = torch.empty(N, E, 4 * H, device=dev)
        ho, co = torch.empty_like(h), torch.empty_like(c)

        in_step = p.encodes_in_step(E, model.compact_obs)      # CommNet on the grid: the observation encoder runs in the launch too
        kw = dict(ob=model.buf_x[0]) if in_step else {}

        def body():
            for _ in range(n):
                p.step_policy_value(enc, h, c, done, pi, act, v, h_out=ho, c_out=co, gates=gates, defer_action_term=True,
                                    mode=ops.SAMPLE_PHILOX, seed=1, env_id_base=0, step=0, **kw)
        # (round 6: the re-step's message term is handed to the next lock-step's policy step -- ONE message product per lock-step
        # is algorithmic work, the second one the reference's two forward passes repeat is not counted when it is not done)
        carry_on = os.environ.get('NMARL_MSG_CARRY', '1') != '0' and p.msg_kind in (ops.MSG_GATHER_RELU, ops.MSG_MEAN_ADD)
        flops = N * E * (2 * (KX + H) * 4 * H + 2 * (2 * H) * 4 * H + (1 if carry_on else 2) * 2 * Km * H + (2 * H * H if in_step else 0))
        # read x (KX - 64 gathered columns), own h, c, the neighbours' h (old, then new) [, the compact observation of self and
        # neighbours]; write h', c', gates, message term, pi, v, action [, the encoder's output]
        nbytes = N * E * ((KX - H + 2 * H) * 4 + 2 * Km * 4 + 2 * H * 4 + 4 * H * 4 + H * 4 + A * 4 + 4 + 1 +
                          ((p.n_obs + H) * 4 if in_step else 0))
        name = 'lstm_step_x_kernel<4,%d,%d,%s> (nmarl_lstm_step_x_msg%s, policy + value of the coupled net in one launch%s%s)' % (
            p.msg_kind, 1 if getattr(p, 'enc_in_kernel', lambda *a: False)(E, model.compact_obs) else 0, '1|2' if carry_on else '0',
            '_grid' if in_step and hasattr(model, '_msg_carry') and N == 25 else '',
            ', observation encoder inside' if in_step else '', ', message term carried between lock-steps' if carry_on else '')
    elif p.can_save_acts:
        # coupled nets: the policy step (kind 1) with the message term computed in its pre-phase where it fits
        KX = p.params[p.k_wx].shape[1]
        enc = p.encode(model.buf_x[0], model.fp)
        pi, act = torch.empty(N, E, A, device=dev), torch.zeros(E, N, dtype=torch.uint8, device=dev)
        gates = torch.empty(N, E, 4 * H, device=dev)
        ho, co = torch.empty_like(h), torch.empty_like(c)
        fused_msg = p._msg() is not None
        Km = p.params['w_msg'].shape[1] if fused_msg else 0

        # lstm_dial: the kernel gathers the senders' message vectors (left by the previous policy step's epilogue) and runs the
        # sender layer on its own new h -- hand it such vectors, so that the timed body is the step kernel alone
        dial = fused_msg and p.msg_kind == ops.MSG_DIAL and getattr(p, '_mfc_img', None) is not None
        mbuf = torch.relu(torch.randn(N, E, H, device=dev)) if dial else None

        def body():
            for _ in range(n):
                if dial:
                    p._m_next = (h.data_ptr(), h._version, mbuf)
                p.step_policy(enc, h, c, done, ho, co, pi, act, gates=gates, mode=ops.SAMPLE_PHILOX, seed=1, env_id_base=0, step=0)
        flops = N * E * (2 * (KX + H) * 4 * H + 2 * Km * H + (2 * H * H if dial else 0))
        nbytes = N * E * ((KX - (H if fused_msg else 0) + 2 * H) * 4 + Km * 4 + 2 * H * 4 + 4 * H * 4 + A * 4 + 1)
        name = 'lstm_step_x_kernel<1,%d> (policy step of the coupled net%s)' % (p.msg_kind if fused_msg else 0,
                                                                              ', in-kernel message term' if fused_msg else '')
    else:
        z = torch.randn(N, E, 4 * H, device=dev)

        def body():
            for _ in range(n):
                ops.lstm_step_fused(h, wh, b, z, None, c, done, None, c, h)
        flops = N * E * 2 * H * 4 * H
        nbytes = N * E * (3 * H + 4 * H + H) * 4
        name = 'lstm_step_mfma16_kernel (nmarl_lstm_step_fused)'
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()                                  # untimed: the first replay also uploads the graph
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n), flops, nbytes, name


def measure_lstm_step_in_rollout(trainer, reps=5):
    """Average duration of an LSTM lock-step launch INSIDE the rollout (what rocprofv3's kernel trace reports for the
    batch), by difference: the n_step rollout captured twice as a hipGraph -- as it runs, and with the LSTM step launches
    left out (same encoders, env steps, bootstrap glue) -- each replayed `reps` times between two HIP events on the
    launch stream; (t_full - t_without) / launches.  Eager event pairs around single launches do not work here: the
    eager rollout is host-bound, the stream idles between launches and the pairs time the host.  State restored after."""
    pol = trainer.model.policy
    names = ['step_policy_value'] if pol.pv_one_launch(trainer.model.E) else ['step_policy', 'step_value']
    n_launch = (trainer.n_step + 1) * len(names)

    def timed(skip):
        orig = {n: getattr(pol, n) for n in names}
        snap = trainer._snapshot()
        try:
            if skip:
                # stubs with the methods' return contracts: step_policy_value / step_policy hand back their action buffer,
                # step_value its value buffer (step_value(enc, h, c, done, h_out, c_out, action, v_out, ...))
                stubs = {'step_policy_value': lambda *a, **k: a[5], 'step_policy': lambda *a, **k: (a[6], a[7]),
                         'step_value': lambda *a, **k: a[7]}
                for n in names:
                    setattr(pol, n, stubs[n])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                trainer._rollout()
        finally:
            for n, f in orig.items():
                setattr(pol, n, f)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        trainer._restore(snap)
        return e0.elapsed_time(e1) * 1e3 / reps

    full, without = timed(False), timed(True)
    return (full - without) / n_launch, full, without


def measure_lstm_step_stamped(trainer):
    """Duration of the lock-step launches inside the rollout from device time stamps: the rollout captured with every lock-step
    launch bracketed by two nmarl_timestamp launches (constant-rate device wall clock), replayed 3 times; the median over all
    launches of (stamp behind - stamp in front).  A duration in the launch's real neighbourhood -- it includes the two kernel
    boundaries next to the stamps (~1.2 us each), which a kernel trace's begin / end do not."""
    from deeprl_network_amd import ops
    pol = trainer.model.policy
    names = ['step_policy_value'] if pol.pv_one_launch(trainer.model.E) else ['step_policy', 'step_value']
    n_launch = (trainer.n_step + 1) * len(names)
    stamps = torch.zeros(n_launch, 2, dtype=torch.int64, device=trainer.device)
    orig = {n: getattr(pol, n) for n in names}
    snap = trainer._snapshot()
    k = [0]

    def bracket(f):
        def g(*a, **kw):
            i = k[0]
            k[0] += 1
            ops.timestamp(stamps[i, 0:1])
            r = f(*a, **kw)
            ops.timestamp(stamps[i, 1:2])
            return r
        return g
    try:
        for n in names:
            setattr(pol, n, bracket(orig[n]))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            trainer._rollout()
    finally:
        for n, f in orig.items():
            setattr(pol, n, f)
    assert k[0] == n_launch, (k[0], n_launch)
    g.replay()
    torch.cuda.synchronize()
    ticks = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        ticks += (stamps[:, 1] - stamps[:, 0]).tolist()
    trainer._restore(snap)
    ticks.sort()
    return ticks[len(ticks) // 2] / ops.timestamp_rate_khz(trainer.device) * 1e3


def bptt_seq_takes_dy(model):
    """The update hands the heads' dL/dh to the one-launch BPTT as dy8 (32 B per row-step, nmarl_lstm_bptt_seq_dy) instead of as a
    [N,T,E,64] tensor (256 B): what agents/models.py `_loss_backward_fused` decides."""
    from deeprl_network_amd import ops
    p = model.policy
    return bool(getattr(p, 'bptt_takes_head_dy', False)) and os.environ.get('NMARL_BPTT_HEAD_DY', '1') != '0' and \
        ops.heads_loss_supported(model.H_all[:, 1:].reshape(model.n_agent, -1, model.n_lstm), model.n_a, p.nbr_idx)


def measure_bptt_seq(model, reps=5):
    """Average duration of the second kernel of the update, the whole reverse recurrence in one launch
    (nmarl_lstm_bptt_seq / _dy), on the model's own saved-activation buffers (shapes [N,T,E,*]): HIP events on the launch
    stream around `reps` launches.  Algorithmic bytes per (agent, replica, step): gates 1 KB + c 256 B read, dz 1 KB written, and
    the heads' dL/dh: 256 B, or 32 B of dy8 when the kernel expands it itself.  Returns (us per launch, bytes per launch)."""
    from deeprl_network_amd import ops
    p = model.policy
    G, C = model.G_buf, model.C_all
    N, T, E, H4 = G.shape
    H = H4 // 4
    G.copy_(torch.rand_like(G))
    C.copy_(torch.randn_like(C) * 0.5)
    dy = bptt_seq_takes_dy(model)
    dHs = None if dy else torch.randn(N, T, E, H, device=G.device)
    head_dy = (torch.randn(N, T * E, 8, device=G.device), torch.randn(N, H, model.n_a + 1, device=G.device)) if dy else None
    dZ = torch.empty_like(G)
    done = torch.zeros(T, E, device=G.device)
    img = ops.lstm_bptt_wimage(None, p.params[p.k_wh])
    for _ in range(2):
        ops.bptt_seq(G, C, done, dHs, img, dZ, head_dy=head_dy)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.bptt_seq(G, C, done, dHs, img, dZ, want_db=False, head_dy=head_dy)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, N * T * E * (H4 + H + (8 if dy else H) + H4) * 4


def measure_bptt_coupled(model, reps=3):
    """Average duration of the coupled nets' reverse recurrence in one launch (nmarl_lstm_bptt_coupled) on the model's own
    saved activations (the trace of the last rollout): HIP events on the launch stream around `reps` calls.  Algorithmic
    bytes per (agent, replica, step): gates 1024 + c 256 + dL/dh 256 [+ relu mask 256] + one message row per real source
    read; dz 1024 + D1 256 + the own message row written.  Returns (us per launch, bytes per launch, kernel name) or None."""
    from deeprl_network_amd import ops
    p = model.policy
    kind = {ops.MSG_GATHER_RELU: ops.COUPLED_NC, ops.MSG_MEAN_ADD: ops.COUPLED_IC3}.get(p.msg_kind)
    H = model.n_lstm
    if kind is None or not ops.bptt_coupled_supported(kind, p.m_max, H):
        return None
    rev = ops.reverse_neighbor_table(p.nbr_idx, kind)
    if rev is None or not ops.bptt_coupled_supported(kind, p.m_max, H, rev=rev):
        return None
    _, wxm, w_msg, _, _, _ = p._seq_args()
    wxm, wh, w_msg = wxm.detach(), p.params[p.k_wh].detach(), w_msg.detach()
    G, C, S = model.G_buf, model.C_all, model.S_buf
    N, T, E, H4 = G.shape
    K = w_msg.shape[1]
    dy = bptt_seq_takes_dy(model)             # the form the update uses: the heads' dL/dh as dy8 (32 B per row-step) or as a tensor (256 B)
    dHs = None if dy else torch.randn(N, T, E, H, device=G.device) * 1e-3
    head_dy = (torch.randn(N, T * E, 8, device=G.device) * 1e-3, torch.randn(N, H, model.n_a + 1, device=G.device) * 0.1) if dy else None
    dZ, D1 = torch.empty_like(G), torch.empty(N, T, E, H, device=G.device)
    done = torch.zeros(T, E, device=G.device)
    ws = (wxm, wh, ops.lstm_bptt_wimage(wxm, wh))
    wm = (w_msg, ops.lstm_bptt_msg_wimage(w_msg))
    mask = S[..., 2 * H:] if kind == ops.COUPLED_NC else None
    ops.bptt_coupled(kind, rev, p.m_max, G, C, done, dHs, ws, wm, mask, dZ, D1, head_dy=head_dy)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.bptt_coupled(kind, rev, p.m_max, G, C, done, dHs, ws, wm, mask, dZ, D1, head_dy=head_dy)
    e1.record()
    torch.cuda.synchronize()
    ops.check_coupled_status()
    sources = float((p.nbr_idx >= 0).sum().item()) / N                    # message rows read per agent (mean fan-in)
    row = 1024 + 256 + (32 if dy else 256) + (256 if mask is not None else 0) + sources * (K // p.m_max if kind == ops.COUPLED_NC else K) * 4 \
        + 1024 + 256 + K * 4
    one = N * -(-E // 128) <= max(_lib_capacity(2, K), 0) and ops.handoff_enabled()
    name = 'lstm_bptt_coupled_kernel<%d,%d,%s,%s> (nmarl_lstm_bptt_coupled%s, %s)' % (
        K // 16, rev['r_row'], 'true' if mask is not None else 'false', 'true' if dy else 'false', ', heads\' dL/dh from dy8' if dy else '',
        'one launch' if one else '%d step-wise launches' % T)
    return e0.elapsed_time(e1) * 1e3 / reps, N * T * E * row, name


def measure_update_graph(trainer, skip=None, stamp=None, reps=5):
    """Duration of the captured update (the hipGraphs BatchedTrainer replays: rewards, return scan, loss, backward, clip + RMSProp,
    epilogue) on the trainer's own buffers: HIP events on the launch stream around `reps` replays of a FRESH capture.
    stamp: name of ONE C-ABI entry point bracketed, inside the capture, by two launches of nmarl_timestamp (a one-thread kernel
    that stores the device's constant-rate wall clock): the DURATION of that launch as it runs inside the update -- clock, cache
    and neighbour-kernel conditions of the real batch -- is the difference of the two stamps (it includes the two kernel
    boundaries next to the stamps, ~3 us).  skip: names of entry points replaced by no-ops during the capture -- the same graph
    without those launches; full - without is their MARGINAL cost (what removing them would buy: not a duration -- taking a
    1.3-kW kernel out hands its power budget to the launches around it).  The trainer's weights, optimiser
    slots, statistics and buffers are put back afterwards.  -> (us per replay, stamped duration in us or None)"""
    from deeprl_network_amd import _lib, ops
    if trainer._upd is None:
        return None, None
    m = trainer.model
    ps = m.policy.params
    state = [t.clone() for t in (ps.flat, ps.ms, trainer.ep_sum, trainer.ep_sq, trainer.ep_len, trainer.fin, m.h_fw, m.c_fw, m.h_bw,
                                 m.c_bw, m.buf_fp[0], m.buf_x[0], trainer.done_pre, m.buf_r)]
    saved = {n: getattr(_lib.lib, n) for n in (skip or ())}
    keep = trainer._upd
    stamps = torch.zeros(2, dtype=torch.int64, device=trainer.device) if stamp else None
    try:
        for n in saved:
            setattr(_lib.lib, n, lambda *a, **k: 0)
        if stamp:
            orig = getattr(_lib.lib, stamp)
            saved[stamp] = orig

            def bracketed(*a, **k):
                ops.timestamp(stamps[0:1])
                rc = orig(*a, **k)
                ops.timestamp(stamps[1:2])
                return rc
            setattr(_lib.lib, stamp, bracketed)
        trainer._upd = None
        trainer._capture_update()
        g = trainer._upd
    finally:
        for n, f in saved.items():
            setattr(_lib.lib, n, f)
        trainer._upd = keep

    def once():
        g['grads'].replay()
        if g['apply'] is not None:
            g['apply'].replay()
    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    inside = []
    e0.record()
    for _ in range(reps):
        once()
        if stamp:
            inside.append(stamps.clone())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    dur = None
    if stamp:
        ticks = [int((x[1] - x[0]).item()) for x in inside]
        dur = sorted(ticks)[len(ticks) // 2] / ops.timestamp_rate_khz(trainer.device) * 1e3      # median, us
    for t, sv in zip((ps.flat, ps.ms, trainer.ep_sum, trainer.ep_sq, trainer.ep_len, trainer.fin, m.h_fw, m.c_fw, m.h_bw, m.c_bw,
                      m.buf_fp[0], m.buf_x[0], trainer.done_pre, m.buf_r), state):
        t.copy_(sv)
    return us, dur


def update_breakdown(trainer, bptt_entry):
    """The captured update timed three ways (5 replays each between two HIP events on the launch stream):
    update_graph_us = as BatchedTrainer replays it; bptt_us_in_update = the DURATION of the reverse-recurrence launch inside it
    (two device time stamps around the launch); marginal_us_in_update = update_graph_us - the same graph without that launch."""
    full, _ = measure_update_graph(trainer)
    if full is None:
        return None
    stamped, dur = measure_update_graph(trainer, stamp=bptt_entry)
    without, _ = measure_update_graph(trainer, skip=(bptt_entry,))
    marginal = full - without
    return {'update_graph_us': full, 'bptt_us_in_update': dur, 'update_graph_us_with_stamps': stamped,
            'update_graph_us_without_bptt': without,
            'marginal_us_in_update': marginal if marginal > 0 else None, 'bptt_entry': bptt_entry,
            'how': 'bptt_us_in_update: the %s launch bracketed by two nmarl_timestamp launches (device wall clock) inside the captured update, '
                   'median of 5 replays -- a duration; marginal_us_in_update: the captured update re-captured without that launch and '
                   'subtracted -- what removing it would buy, not a duration' % bptt_entry}


def _quote_bptt_in_update(rb, ub):
    """roofline_bptt's us_per_launch / achieved / frac on the launch's DURATION inside the captured update (time stamps), the
    back-to-back figure kept beside it."""
    if not isinstance(rb, dict) or 'bytes_per_launch' not in rb or not ub.get('bptt_us_in_update'):
        return
    rb['us_per_launch_back_to_back'] = rb['us_per_launch']
    rb['us_per_launch'] = ub['bptt_us_in_update']
    rb['achieved'] = rb['bytes_per_launch'] / rb['us_per_launch'] / 1e3
    rb['frac'] = rb['achieved'] / HBM_PEAK_GBPS
    rb['marginal_us_in_update'] = ub.get('marginal_us_in_update')
    rb['how'] = rb.get('how', '') + ' | us_per_launch / achieved / frac: the launch\'s duration INSIDE the captured update, two device ' \
        'time stamps (nmarl_timestamp) around it, median of 5 replays; us_per_launch_back_to_back: HIP events around isolated calls; ' \
        'marginal_us_in_update: update graph with - without the launch (not a duration)'


def _lib_capacity(which, K):
    from deeprl_network_amd import _lib
    return _lib.lib.nmarl_handoff_capacity(which, int(K))


OTHER_CONFIGS = ('config_ma2c_nc_slowdown.ini',      # BASELINE.json configs[2]
                 'config_ma2c_cnet_grid.ini',        # configs[3]
                 'config_ma2c_nc_catchup.ini')       # configs[4], one GPU's share (4096 of the 32768 replicas)


def run_other_config(args, cfg_name, device):
    """One of the other BASELINE configs, measured like the headline (fresh env / model / trainer, `--warmup` untimed batches
    -- the first tunes this config's GEMM shapes --, then `--other-steps` timed batches between device synchronisations),
    plus the rooflines of its two hand-written matrix-core kernels on the job's own buffers."""
    import gc
    cp = configparser.ConfigParser()
    cp.read(os.path.join(ROOT, 'config', cfg_name))
    sub = argparse.Namespace(**vars(args))
    sub.envs = 0
    from deeprl_network_amd import ops
    # a fresh job: a time-out of an earlier config in this process must neither pin this one to the slow forms nor make its
    # guarded optimiser steps refuse (both are reported below if they happen here)
    ops.enable_inkernel_handoff()
    ops.handoff_clear(device)
    E, env, model, trainer = make_job(sub, cp, device, 0, 1, None)
    for _ in range(max(2, args.warmup)):
        trainer.run_batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.other_steps):
        trainer.run_batch()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_step, n_agent = model.n_step, env.n_agent
    res = {'workload': '%s, %d agents x %d replicas, %s (%s), n_step %d'
                       % ('ATSC 5x5 grid (synthetic)' if env.name.startswith('atsc') else 'CACC ' + env.name, n_agent, E, env.agent,
                          cfg_name, n_step),
           'steps': args.other_steps, 'ms_per_step': elapsed / args.other_steps * 1e3,
           'value': n_agent * E * n_step * args.other_steps / elapsed, 'unit': 'env-steps/s',
           'a2c_updates_per_s': args.other_steps / elapsed, 'handoff_fallbacks': trainer.handoff_fallbacks,
           'one_launch_lock_step': bool(model.policy.pv_one_launch(E)),
           'hipgraph_update': trainer._upd is not None, 'update_capture_error': trainer.update_capture_error,
           'inkernel_handoff_enabled': bool(ops.handoff_enabled())}
    try:
        us_l, flops_l, bytes_l, lname = measure_lstm_step(model)
        us_iso = us_l
        try:
            us_l = measure_lstm_step_in_rollout(trainer)[0]
        except Exception as ex:
            res['roofline_in_rollout_error'] = repr(ex)
        try:
            us_st = measure_lstm_step_stamped(trainer)
        except Exception as ex:
            us_st = None
            res['roofline_stamped_error'] = repr(ex)
        res['roofline'] = {'kernel': lname, 'bound': 'mfma', 'us_per_launch': us_l, 'us_per_launch_isolated_graph': us_iso,
                           'us_per_launch_stamped_in_rollout': us_st,
                           'frac_stamped': None if not us_st else flops_l / us_st / 1e6 / MFMA_F32_PEAK_TFLOPS,
                           'achieved': flops_l / us_l / 1e6, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': flops_l / us_l / 1e6 / MFMA_F32_PEAK_TFLOPS, 'flops_per_launch': flops_l,
                           'launches_per_batch': (n_step + 1) * (1 if model.policy.pv_one_launch(E) else 2),
                           'how': 'as the headline `roofline`: in-rollout launch time by difference of two hipGraph timings'}
    except Exception as ex:
        res['roofline'] = {'error': repr(ex)}
    try:
        m = measure_bptt_coupled(model)
        if m is not None:
            us_b, bytes_b, bname = m
            res['roofline_bptt'] = {'kernel': bname, 'bound': 'hbm', 'us_per_launch': us_b, 'achieved': bytes_b / us_b / 1e3,
                                    'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': bytes_b / us_b / 1e3 / HBM_PEAK_GBPS,
                                    'bytes_per_launch': bytes_b, 'launches_per_batch': 1}
    except Exception as ex:
        res['roofline_bptt'] = {'error': repr(ex)}
    try:
        ub = update_breakdown(trainer, 'nmarl_lstm_bptt_coupled')
        if ub is not None:
            res['update'] = ub
            _quote_bptt_in_update(res.get('roofline_bptt'), ub)
    except Exception as ex:
        res['update'] = {'error': repr(ex)}
    del trainer, model, env
    gc.collect()
    torch.cuda.empty_cache()
    return res


def pmc_traffic(key):
    """HBM bytes per replica-step from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE runs of tools/pmc_env.py, FETCH x2 per MI355X_MICROARCH.md, calibrated on a known copy).
    PMC counters cannot be read from inside this process, so a committed measurement is reported: for every key the one
    of the HIGHEST round (the number in the file name) that holds it -- a later round need not re-measure kernels it did
    not touch."""
    import glob
    import re
    best = (None, None, -1)
    for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')):
        m = re.match(r'r(\d+)[a-z]?_pmc_traffic\.json$', os.path.basename(f))
        if not m or int(m.group(1)) <= best[2]:
            continue
        k = json.load(open(f)).get('kernels', {}).get(key)
        if k:
            best = (k['traffic_bytes_per_replica'], os.path.basename(f), int(m.group(1)))
    return best[0], best[1]


def cpu_port_worker(cfg_path, n_batches):
    """`bench.py --cpu-port-worker N`: one replica of the restated E=1 CPU loop on one thread; prints steps, seconds, agents."""
    from oracle import trainer_ref
    torch.set_num_threads(1)
    cp = configparser.ConfigParser()
    cp.read(cfg_path)
    env, model, tr = trainer_ref.build(cp)
    tr.run_batches(1)
    steps, sec = tr.run_batches(n_batches)
    print('CPUPORT %d %.6f %d' % (steps, sec, env.n_agent))


def _host_cpu():
    model = 'unknown'
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                model = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return model, ncores


def _time_port(cfg_path, n_batches):
    from oracle import trainer_ref
    cp = configparser.ConfigParser()
    cp.read(cfg_path)
    env, model, tr = trainer_ref.build(cp)
    tr.run_batches(1)                                   # warm-up (allocator, first-touch)
    steps, sec = tr.run_batches(n_batches)
    return env, steps, sec


BASELINE_CONFIG0 = os.path.join(ROOT, 'config', 'config_ia2c_catchup.ini')     # BASELINE.json configs[0]


def cpu_baseline(cfg_path, n_batches):
    """Reference-equivalent E=1 CPU loop (restated; TF-1.12 cannot run here), timed on THIS host's cores:
    `value` = BASELINE.json configs[0] to the letter -- CACC catch-up, 8 agents, 1 env, IA2C (config_ia2c_catchup.ini) -- on
    ONE core, whatever --config the GPU side ran; `all_cores` = the same as one independent replica per core (8 processes);
    `workload_matched` = the port of the GPU workload's own algorithm (--config) on one core; plus the env-only numbers of
    the REAL reference env (a committed profile: the reference checkout does not exist on the GPU box)."""
    import subprocess
    torch.set_num_threads(1)
    cpu_model, ncores = _host_cpu()
    env, steps, sec = _time_port(BASELINE_CONFIG0, n_batches)
    out = {'value': steps * env.n_agent / sec, 'unit': 'env-steps/s (agents x envs x steps/s)', 'cores': 1,
           'kind': 'port', 'config': 'BASELINE configs[0]: ' + os.path.basename(BASELINE_CONFIG0),
           'host_cpu': cpu_model, 'host_cores': ncores,
           'sample': '%d n_step batches (%d env steps, E=1, IA2C catch-up) of the restated reference loop '
                     '(oracle/trainer_ref.py: NumPy env + per-agent torch-CPU LSTMs + TF-RMSProp), %.1f s'
                     % (n_batches, steps, sec),
           'updates_per_s': n_batches / sec}
    try:        # the same loop as one independent replica (process) per host core
        nproc = min(ncores, 8)                  # 8 processes (a fresh box imports torch slowly per process)
        nb = max(10, n_batches // 4)
        envv = dict(os.environ, OMP_NUM_THREADS='1', HIP_VISIBLE_DEVICES='', NMARL_BENCH_TUNABLEOP='0')
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-port-worker', str(nb), '--config', BASELINE_CONFIG0],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=envv) for _ in range(nproc)]
        res = []
        for p_ in ps:
            o, _ = p_.communicate(timeout=240)
            ln = [x for x in o.splitlines() if x.startswith('CPUPORT')][0].split()
            res.append((int(ln[1]), float(ln[2]), int(ln[3])))
        out['all_cores'] = {'value': sum(s_ * n / t for s_, t, n in res), 'cores': nproc, 'host_cores': ncores, 'kind': 'port',
                            'sample': '%d processes x %d batches, %.1f s wall' % (nproc, nb, time.perf_counter() - t0)}
    except Exception as ex:
        out['all_cores'] = {'error': repr(ex)}
        for p_ in locals().get('ps', []):      # never leave our own workers behind
            if p_.poll() is None:
                p_.kill()
    if os.path.abspath(cfg_path) != os.path.abspath(BASELINE_CONFIG0):
        try:
            nb = max(10, n_batches // 2)
            env_w, steps_w, sec_w = _time_port(cfg_path, nb)
            out['workload_matched'] = {'value': steps_w * env_w.n_agent / sec_w, 'cores': 1, 'kind': 'port',
                                       'config': os.path.basename(cfg_path),
                                       'sample': '%d batches (%d env steps, E=1, %s) of the same port, %.1f s'
                                                 % (nb, steps_w, env_w.agent, sec_w)}
            env = env_w
        except Exception as ex:
            out['workload_matched'] = {'error': repr(ex)}
    import glob
    refs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_cpu_env_reference.json')))      # the latest round's
    ref = refs[-1] if refs else ''
    if os.path.exists(ref):
        d = json.load(open(ref))
        key = '%s_%s' % (env.agent, env.name)
        if key in d['runs']:
            out['reference_env_only'] = dict(d['runs'][key], kind='reference', unit=d['unit'], cpu_model=d['cpu_model'],
                                             what=d['what'] + '; measured in the authoring container by '
                                             'tools/cpu_env_baseline.py (the reference checkout does not exist on the GPU box)')
    return out


def _ordered(out):
    """The JSON line with the contract's objects up front (a reader that keeps only the head of the line still sees the
    headline, `roofline`, `roofline_bptt` and `cpu_baseline`): the long `how` texts move to ONE `how` object at the end, the
    contract keys of each object come first."""
    how = {}

    def strip(key, d):
        if isinstance(d, dict):
            if isinstance(d.get('how'), str):
                how[key] = d.pop('how')
            for k, v in list(d.items()):
                strip(key + '.' + k, v)
        elif isinstance(d, list):
            for i, v in enumerate(d):
                strip('%s[%d]' % (key, i), v)
    for k, v in out.items():
        strip(k, v)

    def lead(d, keys):
        if not isinstance(d, dict):
            return d
        return {**{k: d[k] for k in keys if k in d}, **{k: v for k, v in d.items() if k not in keys}}
    roof = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'us_per_launch')
    for k in list(out):
        if k.startswith('roofline'):
            out[k] = lead(out[k], roof)
    if 'cpu_baseline' in out:
        out['cpu_baseline'] = lead(out['cpu_baseline'], ('value', 'unit', 'cores', 'kind', 'sample'))
    first = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
             'dtype', 'data', 'config', 'roofline', 'roofline_bptt', 'cpu_baseline')
    res = lead(out, first)
    res['how'] = how
    return res


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: re-exec this script under
    torch.distributed.run with N ranks on this node (one rank per GPU, rendezvous on 127.0.0.1)
    and pass its exit code through.  Rank 0 of the child job prints the ONE JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def make_job(args, cp, device, rank, world, group):
    """(env, model, trainer) of one rank: E replicas with global replica ids rank*E ... (Philox streams)."""
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs import make_batch_env
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    E = args.envs or cp['ENV_CONFIG'].getint('num_envs', fallback=4096)
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=E, device=device, env_id_base=rank * E)
    np.random.seed(env.seed)                               # identical initial weights on every rank
    cls = {'ia2c': models.IA2C, 'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3,
           'ma2c_cu': models.IA2C_CU, 'ma2c_dial': models.MA2C_DIAL}[env.agent]
    model = cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, int(1e9),
                cp['MODEL_CONFIG'], seed=env.seed, num_envs=E, device=device, dist_group=group,
                n_feat_ls=getattr(env, 'n_feat_ls', None))
    trainer = BatchedTrainer(env, model, Counter(int(1e18), int(1e18), int(1e18)), use_graph=not args.no_graph,
                             rank=rank, world_size=world)
    return E, env, model, trainer


def tune_once(args, rank):
    """N > 1: the library GEMMs are tuned ONCE.  Rank 0 runs a throw-away single-GPU job of the same shapes in a child
    process (`--tune-only`: one batch, no collective), whose TunableOp results land in one shared CSV
    (PYTORCH_TUNABLEOP_FILENAME); the other ranks wait at a barrier, then every rank reads that file and switches
    tuning off -- instead of N ranks tuning the same shapes concurrently inside --warmup."""
    import subprocess
    import torch.cuda.tunable as tunable
    import torch.distributed as dist
    if not tunable.is_enabled():
        return
    path = os.environ.get('NMARL_TUNABLEOP_SHARED', '/tmp/nmarl_tunableop_shared_%d.csv' % os.getppid())
    if rank == 0:
        drop = ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'GROUP_WORLD_SIZE', 'ROLE_RANK',
                'ROLE_WORLD_SIZE', 'ROLE_NAME', 'NMARL_BENCH_FORCE_DIST')
        env = {k: v for k, v in os.environ.items()
               if k not in drop and not k.startswith('TORCHELASTIC') and not k.startswith('MASTER_')}
        env['PYTORCH_TUNABLEOP_FILENAME'] = path
        cmd = [sys.executable, os.path.abspath(__file__), '--tune-only', '--gpus', '1', '--config', args.config,
               '--envs', str(args.envs)] + (['--no-graph'] if args.no_graph else [])
        subprocess.call(cmd, env=env, stdout=subprocess.DEVNULL)
    dist.barrier()
    if os.path.exists(path):
        tunable.read_file(path)
    tunable.tuning_enable(False)                           # every rank replays the one set of choices
    dist.barrier()


def main():
    args = parse()
    if args.cpu_port_worker:
        return cpu_port_worker(args.config, args.cpu_port_worker)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'RANK' not in os.environ and args.gpus > 1:
        self_launch(args)
    if args.gpus != world:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    import torch.distributed as dist
    # test hooks (1-GPU boxes): NMARL_BENCH_ONE_DEVICE=1 maps every rank to cuda:0, NMARL_DIST_BACKEND=gloo swaps RCCL,
    # NMARL_BENCH_FORCE_DIST=1 creates the process group (and the gradient all-reduce) even for one rank
    if os.environ.get('NMARL_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
        if world > 1:       # ranks sharing a device: their blocks are not co-resident, no in-launch hand-off (ops.step_handoff_supported)
            os.environ['NMARL_INKERNEL_HANDOFF'] = '0'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    group = None
    use_dist = world > 1 or os.environ.get('NMARL_BENCH_FORCE_DIST') == '1'
    if use_dist:
        backend = os.environ.get('NMARL_DIST_BACKEND', 'nccl')   # 'nccl' IS RCCL on ROCm (xGMI)
        if 'RANK' not in os.environ:
            os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
                              MASTER_PORT=os.environ.get('MASTER_PORT', '29611'))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    cp = configparser.ConfigParser()
    cp.read(args.config)
    if world > 1:
        tune_once(args, rank)
    from deeprl_network_amd.envs import make_batch_env
    E, env, model, trainer = make_job(args, cp, device, rank, world, group)
    if args.tune_only:                                     # child of tune_once: one batch tunes every GEMM shape
        trainer.run_batch()
        torch.cuda.synchronize()
        return
    for _ in range(args.warmup):
        trainer.run_batch()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.run_batch()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms, allreduce_us = [elapsed / args.steps * 1e3], None
    if use_dist:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        elapsed = max(float(x.item()) for x in every)                  # MAX over ranks
        # the path's one collective, timed on its own after the timed region: the flat gradient all-reduce of an update
        g = model.policy.params.grad_wire
        for _ in range(3):
            dist.all_reduce(g, group=group)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(g, group=group)
        torch.cuda.synchronize()
        allreduce_us = (time.perf_counter() - t1) / 20 * 1e6

    n_step = model.n_step
    is_grid = env.name.startswith('atsc')
    n_agent = env.n_agent
    # algorithmic bytes per replica-step of the env kernel
    # grid: read q, transit 1200 + action 25 + prev 25 + t 4 + xi 16; write q, transit 1200 + prev 25 + t 4 + reward /
    # global reward / done 9 + observation (compact [25,12] 1200 B; gathered slab: 5040 B of its 6000 are data)
    balg = (3708 if trainer.compact_obs else 7548) if is_grid else (B_ALG_COMPACT if trainer.compact_obs else B_ALG_GATHERED)
    if is_grid:
        obs_variant = 'compact [E,25,12] observation' if trainer.compact_obs else 'gathered [E,25,60] observation'
    else:
        obs_variant = 'compact [E,8,5] observation' if trainer.compact_obs else 'gathered [E,8,15] observation'
    kname = 'grid_step_kernel (nmarl_grid_step)' if is_grid else 'cacc_step_kernel (nmarl_cacc_step)'
    if env.name.endswith('real_net'):
        # q, transit in and out (4 B x 2 x 2 x 264 links) + action / prev bytes + scalars + the padded neighbour slab
        tp = env.topo
        balg = 16 * sum(tp.n_s_ls) + 3 * tp.N + 24 + 4 * tp.L * (1 + tp.m_max) * tp.N
        kname = 'net_step_kernel (nmarl_net_step)'
    env_steps = n_agent * E * n_step * args.steps * world
    out = {
        'metric': 'env-steps/sec (agents x envs x steps/s), full rollout + A2C update loop',
        'value': env_steps / elapsed, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'a2c_updates_per_s': args.steps / elapsed,
        'per_rank_ms_per_step': rank_ms,
        'grad_allreduce': None if allreduce_us is None else {
            'us': allreduce_us, 'bytes': int(model.policy.params.grad_wire.numel()) * 4,
            'per_update': getattr(model, 'allreduce_calls', 0) / max(1, trainer.n_batches),      # counted: collectives issued / updates run
            'backend': os.environ.get('NMARL_DIST_BACKEND', 'nccl'),
            'how': '20 back-to-back all_reduce(sum) of the flat [N,P] fp32 gradient after the timed region, host clock around '
                   'them with a device synchronize on both sides (mean; the timed region contains exactly one per step)'},
        'lock_steps_per_s': n_step * args.steps / elapsed,
        'config': {'workload': '%s, %d agents x %d replicas/GPU, %s (%s), n_step %d'
                               % (('ATSC Monaco-like network (synthetic, heterogeneous agents)' if env.name.endswith('real_net') else 'ATSC 5x5 grid (synthetic)') if is_grid else 'CACC ' + env.name, n_agent, E, env.agent,
                                  os.path.basename(args.config), n_step),
                   'replicas_per_gpu': E, 'global_replicas': E * world, 'parallelism': 'dp%d' % world,
                   'hipgraph_rollout': trainer.use_graph, 'hipgraph_update': trainer._upd is not None,
                   'update_capture_error': trainer.update_capture_error,
                   'gemm_autotune': os.environ.get('PYTORCH_TUNABLEOP_ENABLED', '0') == '1',
                   'step_definition': 'one n_step batch: %d lock-steps (2 LSTM steps each, quirk Q1) + bootstrap + '
                                      '1 A2C update over all replicas' % n_step},
    }
    if rank == 0:
        # ---- `roofline`: the DOMINANT kernel of the batch = the fused MFMA LSTM lock-step (fp32 matrix pipe)
        if model.n_lstm == 64:
            try:
                us_l, flops_l, bytes_l, lname = measure_lstm_step(model)
                x_side = model.policy.can_save_acts
                lpb = (n_step + 1) if model.policy.fused_pv else 2 * (n_step + 1)
                # (PMC key of the kernel the rollout launches: round 5's one-launch lock-step has its own measurement)
                lkey = 'lstm_step_x_enc_env_N8_E4096' if getattr(trainer, 'env_in_kernel', False) else 'lstm_step_x_N8_E4096'
                us_iso = us_l
                us_roll = None
                if x_side:
                    try:          # the figure the roofline is quoted on: the launch as it runs inside the rollout
                        us_roll, us_roll_full, us_roll_without = measure_lstm_step_in_rollout(trainer)
                        us_l = us_roll
                    except Exception as ex:
                        us_roll = None
                        out['roofline_in_rollout_error'] = repr(ex)
                us_stamp = None
                if x_side:
                    try:
                        us_stamp = measure_lstm_step_stamped(trainer)
                    except Exception as ex:
                        out['roofline_stamped_error'] = repr(ex)
                ach = flops_l / us_l / 1e6
                out['roofline'] = {
                    'kernel': lname, 'bound': 'mfma' if x_side else 'hbm',
                    'achieved': ach if x_side else bytes_l / us_l / 1e3,
                    'peak': MFMA_F32_PEAK_TFLOPS if x_side else HBM_PEAK_GBPS,
                    'unit': 'TFLOP/s' if x_side else 'GB/s',
                    'frac': ach / MFMA_F32_PEAK_TFLOPS if x_side else bytes_l / us_l / 1e3 / HBM_PEAK_GBPS,
                    'traffic': (lambda t: None if (t[0] is None or not x_side or n_agent * E != 8 * 4096) else t[0] * n_agent * E)(
                        pmc_traffic(lkey)),
                    'traffic_source': pmc_traffic(lkey)[1], 'flops_per_launch': flops_l, 'bytes_per_launch': bytes_l, 'us_per_launch': us_l,
                    'us_per_launch_isolated_graph': us_iso, 'us_per_launch_in_rollout': us_roll,
                    'us_per_launch_stamped_in_rollout': us_stamp,
                    'frac_stamped': None if not us_stamp else flops_l / us_stamp / 1e6 / MFMA_F32_PEAK_TFLOPS,
                    'rollout_graph_us': None if us_roll is None else us_roll_full,
                    'rollout_graph_us_without_lstm_steps': None if us_roll is None else us_roll_without,
                    'frac_isolated_graph': flops_l / us_iso / 1e6 / MFMA_F32_PEAK_TFLOPS if x_side else None,
                    'rows_per_launch': n_agent * E, 'launches_per_batch': lpb,
                    'hbm_frac_of_same_launch': bytes_l / us_l / 1e3 / HBM_PEAK_GBPS,
                    'how': 'achieved / frac use the launch duration INSIDE the rollout, by difference: the n_step rollout captured as a '
                           'hipGraph with and without its LSTM lock-step launches, 5 replays each between two HIP events on the launch '
                           'stream, (t_full - t_without) / launches (agrees with the rocprofv3 kernel trace of the batch, profiles/).  '
                           'us_per_launch_stamped_in_rollout / frac_stamped: the same launches bracketed by two nmarl_timestamp launches each '
                           '(device wall clock) inside the captured rollout, median of 3 x 61 -- a duration that includes the two kernel '
                           'boundaries next to the stamps.  '
                           'us_per_launch_isolated_graph: hipGraph of 60 back-to-back launches on the model shapes and weights (hot caches: '
                           'flatters the kernel by 5-8 %%).  Algorithmic work per (agent, replica) row: '
                           'policy step 2*(KX+64)*256 flops + value re-step 2*64*256 flops (uncoupled nets; coupled nets: the '
                           'policy step + its 2*K_m*64 message flops, the value step is a second launch), fp32 in / fp32 accumulate on '
                           'v_mfma_f32_16x16x4_f32 (peak %.1f TFLOP/s dense, MI355X_MICROARCH.md); the same launch moves '
                           '%.1f MB of algorithmic HBM bytes (x, h, c in; h, c, gates, pi, v, action out), i.e. it is '
                           'matrix-pipe-bound, not HBM-bound' % (MFMA_F32_PEAK_TFLOPS, bytes_l / 1e6)}
                if getattr(trainer, 'enc_in_kernel', False) and x_side:
                    # round 5: this launch also runs both input encoders (matrix-core pre-phase) and, with env_in_kernel, the CACC env
                    # step -- the work of round 4's SECOND launch per lock-step (cacc_step_encode_kernel, ~10.9 us).  `frac` keeps
                    # counting the LSTM flops only (comparable with earlier rounds: 48.8 us = 0.56 then); the fields below say what
                    # else the duration pays for.  Same-box A/B of the three forms: profiles/r05_ab_lockstep.txt.
                    enc_fl = n_agent * E * 2 * 23 * 64
                    out['roofline']['absorbed_work'] = {
                        'input_encoders_flops_per_launch': enc_fl,
                        'frac_with_encoder_flops': (flops_l + enc_fl) / us_l / 1e6 / MFMA_F32_PEAK_TFLOPS,
                        'env_step_inside': bool(getattr(trainer, 'env_in_kernel', False)),
                        'launches_per_lock_step': 1 if getattr(trainer, 'env_in_kernel', False) else 2,
                        'note': 'the launch replaces lstm_step_x_kernel<3,0,0> + cacc_step_encode_kernel of round 4; a rollout of n_step '
                                'lock-steps is n_step + 1 of these launches and nothing else but the bootstrap bookkeeping '
                                '(rollout_graph_us_without_lstm_steps)'}
            except Exception as ex:
                out['roofline'] = {'error': repr(ex)}
        # ---- the update's recurrence (uncoupled nets with saved activations): one HBM-bound launch
        if getattr(model, 'save_acts', False) and not model.policy.coupled and model.n_lstm == 64:
            try:
                us_b, bytes_b = measure_bptt_seq(model)
                dy_form = bptt_seq_takes_dy(model)
                out['roofline_bptt'] = {
                    'kernel': 'lstm_bptt_seq_kernel<%s> (%s: %d reverse steps in one launch)' %
                              ('true' if dy_form else 'false', 'nmarl_lstm_bptt_seq_dy' if dy_form else 'nmarl_lstm_bptt_seq', n_step),
                    'bound': 'hbm', 'achieved': bytes_b / us_b / 1e3, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                    'frac': bytes_b / us_b / 1e3 / HBM_PEAK_GBPS,
                    'traffic': (lambda t: None if (t[0] is None or n_agent * E * n_step != 8 * 4096 * 60) else t[0] * n_agent * E * n_step)(
                        pmc_traffic('lstm_bptt_seq_N8_E4096_T60')),
                    'traffic_source': pmc_traffic('lstm_bptt_seq_N8_E4096_T60')[1], 'bytes_per_launch': bytes_b,
                    'us_per_launch': us_b, 'launches_per_batch': 1,
                    'how': 'HIP events around 5 launches on the model\'s own [N,T,E,*] buffers; algorithmic bytes per (agent, '
                           'replica, step) = gates 1024 + c 256 + %s read + dz 1024 written = %d B' %
                           (('dy8 32 (the heads\' dL/dh is expanded inside the kernel)', 2336) if dy_form else ('dL/dh 256', 2560))}
            except Exception as ex:
                out['roofline_bptt'] = {'error': repr(ex)}
            try:
                ub = update_breakdown(trainer, 'nmarl_lstm_bptt_seq_dy' if bptt_seq_takes_dy(model) else 'nmarl_lstm_bptt_seq')
                if ub is not None:
                    out['update'] = ub
                    _quote_bptt_in_update(out['roofline_bptt'], ub)
            except Exception as ex:
                out['update'] = {'error': repr(ex)}
        # ---- the env-step kernel (north_star's HBM roofline), measured live on this rank's stream
        tape = model.buf_act.clone()
        us = measure_step_kernel(env, tape)
        ach = balg * E / us / 1e3
        pk = 'cacc_step_compact' if trainer.compact_obs else 'cacc_step'
        tr_small, tr_src = pmc_traffic(pk + '_E4096') if (not is_grid and E == 4096) else (None, None)
        out['roofline_env_step'] = {
            'kernel': kname, 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS, 'traffic': None if tr_small is None else tr_small * E,
            'traffic_source': tr_src,
            'bytes_per_launch': balg * E, 'us_per_launch': us, 'replicas_per_launch': E,
            'how': 'hipGraph of %d back-to-back step launches on the rollout state with the batch action tape, '
                   '20 replays between two HIP events on the launch stream (includes graph-node gaps). '
                   'B_alg = %d B/replica-step (%s). At E=%d the launch moves %.2f MB: '
                   'latency-bound and LLC-resident (SURVEY.md H1); see roofline_env_step_large_E for the HBM regime.'
                   % (n_step, balg, obs_variant, E, balg * E / 1e6)}
        if getattr(trainer, 'fused_encode', False):
            # what the rollout actually launches per lock-step: the step AND the next lock-step's encoders (csrc/cacc.hip
            # cacc_step_encode_kernel).  Algorithmic bytes per replica-step: the env's 351 + fingerprints read 8 x 16 + the
            # encoded LSTM input written 8 agents x 128 floats
            try:
                spec = model.policy.fused_env_encode(model.buf_fp[1], model.encode_target(1))
                us_f = measure_step_kernel(env, tape, encode=spec)
                bf = B_ALG_COMPACT + 8 * 16 + 8 * 128 * 4
                tr_f, tr_fs = pmc_traffic('cacc_step_encode_E4096') if E == 4096 else (None, None)
                out['roofline_env_step_fused'] = {
                    'kernel': 'cacc_step_encode_kernel (nmarl_cacc_step_encode: env step + next lock-step input encoders)',
                    'bound': 'hbm', 'achieved': bf * E / us_f / 1e3, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                    'frac': bf * E / us_f / 1e3 / HBM_PEAK_GBPS, 'traffic': None if tr_f is None else tr_f * E, 'traffic_source': tr_fs,
                    'bytes_per_launch': bf * E, 'us_per_launch': us_f, 'replicas_per_launch': E,
                    'how': 'hipGraph of %d back-to-back launches on the rollout state, 20 replays between two HIP events; B_alg = '
                           '%d B/replica-step = env 351 + fingerprints 128 + encoded LSTM input 4096.  Latency-bound at this size like the '
                           'plain step (%.1f MB per launch)' % (n_step, bf, bf * E / 1e6)}
            except Exception as ex:
                out['roofline_env_step_fused'] = {'error': repr(ex)}
        if 'roofline' not in out or 'error' in out['roofline']:
            if 'roofline' in out:
                out['roofline_lstm_error'] = out['roofline']['error']
            out['roofline'] = dict(out['roofline_env_step'])
        if world == 1 and not env.name.endswith('real_net'):
            try:
                big_E = (1 << 17) if is_grid else (1 << 21)
                big = make_batch_env(cp['ENV_CONFIG'], num_envs=big_E, device=device, env_id_base=10 ** 7)
                if trainer.compact_obs:
                    big.set_compact_obs(True)
                big.reset()
                e = torch.arange(big_E, device=device)[:, None]
                a = torch.arange(n_agent, device=device)[None, :]
                big_tape = torch.stack([((e + 3 * a + s) % env.n_a).to(torch.uint8) for s in range(8)])
                for k in range(60):                      # leave the all-equilibrium start of the episode
                    big.step(big_tape[k % 8], auto_reset=True)
                us_b = measure_step_kernel(big, big_tape, reps=10)
                ach_b = balg * big_E / us_b / 1e3
                tr_big, tr_src_b = pmc_traffic(('grid_step_compact_E2p17' if trainer.compact_obs else 'grid_step_E2p17') if is_grid else pk + '_E2p21')
                kname_big = kname if (is_grid or not trainer.compact_obs or os.environ.get('NMARL_CACC_QUAD', '1') == '0') else \
                    'cacc_step4_kernel (nmarl_cacc_step in the HBM regime: four vehicles per lane, 16-byte accesses)'
                out['roofline_env_step_large_E'] = {'kernel': kname_big, 'bound': 'hbm', 'achieved': ach_b,
                                                    'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach_b / HBM_PEAK_GBPS,
                                                    'traffic': None if tr_big is None else tr_big * big_E,
                                                    'traffic_source': tr_src_b,
                                                    'replicas_per_launch': big_E, 'us_per_launch': us_b,
                                                    'bytes_per_launch': balg * big_E,
                                                    'how': 'same kernel (' + obs_variant + ') at E=2^%d (working set >> 256 MB Infinity Cache), '
                                                           'actions (env+3*agent+step) mod n_a (SURVEY.md 8d)' % (17 if is_grid else 21)}
                del big
            except Exception as ex:      # never lose the headline line to the side measurement
                out['roofline_env_step_large_E'] = {'error': repr(ex)}
        if world == 1 and not args.no_cpu_baseline and not is_grid:
            out['cpu_baseline'] = cpu_baseline(args.config, args.cpu_batches)
        # ---- the other BASELINE configs (coupled nets), measured the same way after the headline's timed region
        default_cfg = os.path.abspath(args.config) == os.path.abspath(os.path.join(ROOT, 'config', 'config_ia2c_fp_catchup.ini'))
        if world == 1 and default_cfg and not args.no_other_configs and not args.envs:
            del trainer, model, env
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out['other_configs'] = []
            for cfg_name in OTHER_CONFIGS:
                try:
                    out['other_configs'].append(run_other_config(args, cfg_name, device))
                except Exception as ex:      # never lose the headline line to a side measurement
                    out['other_configs'].append({'workload': cfg_name, 'error': repr(ex)})
        print(json.dumps(_ordered(out)))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
