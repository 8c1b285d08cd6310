"""torch-CPU restatements of every HIP op of include/nmarl.h (neighbour
aggregation, LSTM cell, action draw, n-step return, clip + TF-RMSProp).
TEST INFRASTRUCTURE: the GPU parity tests compare each kernel against these,
and the CPU test-suite patches them into `deeprl_network_amd.ops` (see
tests/cpu_emulation.py) to exercise the host logic without a GPU.  The product
never imports this module.

Each function follows the reference lines cited in include/nmarl.h.
"""
import numpy as np
import torch


def neighbor_lists(nbr_idx):
    tab = nbr_idx.cpu().numpy()
    return [[int(j) for j in row if j >= 0] for row in tab]


def nbr_gather(x, nbr_idx):
    """boolean_mask + reshape (agents/utils.py:192-195): [N,E,F] -> [N,E,m_max*F]."""
    N, E, F = x.shape
    m = nbr_idx.shape[1]
    out = []
    for i, js in enumerate(neighbor_lists(nbr_idx)):
        parts = [x[j] for j in js] + [torch.zeros_like(x[0])] * (m - len(js))
        out.append(torch.cat(parts, dim=-1))
    return torch.stack(out, dim=0)


def nbr_mean(x, nbr_idx):
    """reduce_mean(boolean_mask(out_m, masks[i])) (agents/utils.py:395)."""
    # an agent without neighbours receives no message (lstm_ic3_hetero, agents/utils.py:481-493: `if n_m:`) -> 0
    return torch.stack([torch.stack([x[j] for j in js], 0).mean(0) if len(js) else torch.zeros_like(x[0]) * x[0]
                        for js in neighbor_lists(nbr_idx)], 0)


def nbr_gather_bwd(dy, nbr_idx, F, add=None):
    """Adjoint of nbr_gather (by autograd of the forward restatement)."""
    N, E, _ = dy.shape
    x = torch.zeros(N, E, F, dtype=dy.dtype, requires_grad=True)
    with torch.enable_grad():
        y = nbr_gather(x, nbr_idx)
    g = torch.autograd.grad(y, x, dy)[0]
    return g if add is None else g + add


def nbr_mean_bwd(dy, nbr_idx, add=None):
    x = torch.zeros_like(dy, requires_grad=True)
    with torch.enable_grad():
        y = nbr_mean(x, nbr_idx)
    g = torch.autograd.grad(y, x, dy)[0]
    return g if add is None else g + add


def cell_bwd(gates, c_prev, c_new, done, dh, dc, dz, dc_prev, dh2=None):
    """Backward of the LSTM cell from the saved post-activation gates (agents/utils.py:102-113)."""
    H = c_prev.shape[-1]
    gi, gf, go, gu = gates.split(H, dim=-1)
    keep = (1.0 - done).view(1, -1, 1)
    tc = torch.tanh(c_new)
    g_h = torch.zeros_like(c_new) if dh is None else dh
    if dh2 is not None:
        g_h = g_h + dh2
    g_c = (torch.zeros_like(c_new) if dc is None else dc) + g_h * go * (1 - tc * tc)
    dz.copy_(torch.cat([g_c * gu * gi * (1 - gi), g_c * (c_prev * keep) * gf * (1 - gf), g_h * tc * go * (1 - go),
                        g_c * gi * (1 - gu * gu)], dim=-1))
    dc_prev.copy_(g_c * gf * keep)


def bptt_supported(H):
    return H == 64


def lstm_bptt_wimage(wxm, wh, out=None):
    return wh if out is None else out     # kernel-side layout in the product; the restatement hands wh through


def bptt_step_db_parts(N, E, H, device):
    return torch.zeros(N, 1, 4 * H)


def bptt_step(gates, c_prev, c_new, done, dh, dh2, dc, ws, dz, dc_prev, dhd, apply_keep, dx=None, mask=None, db_part=None):
    """One reverse step of the unrolled LSTM graph: cell backward, then [dx | dhd] = dz @ [wxm; wh]^T; db_part += column
    sums of dz (the bias gradient's share of this step)."""
    cell_bwd(gates, c_prev, c_new, done, dh, dc, dz, dc_prev, dh2=dh2)
    if db_part is not None:
        db_part[:, 0] += dz.sum(dim=1).to(db_part.dtype)
    wxm, wh, _ = ws
    r = torch.bmm(dz, wh.transpose(1, 2))
    if apply_keep:
        r = r * (1.0 - done).view(1, -1, 1)
    dhd.copy_(r)
    if wxm is not None:
        v = torch.bmm(dz, wxm.transpose(1, 2))
        if mask is not None:
            v = v * (mask > 0)
        dx.copy_(v)


BPTT_SEQ_MAX_E = 1 << 21


def bptt_seq(G, Call, done, dHs, img, dZ, want_db=True, want_state_grad=False, wh=None, head_dy=None):
    """T reverse steps of bptt_step (KM = 0, every step masked) -> (db, dh0, dc0); wh is the restatement's weight
    operand (the product passes the kernel-side image `img`, which the restatement cannot read back).
    head_dy = (dy8, hw): the heads' dL/dh = dy hw^T, as in bptt_coupled."""
    N, T, E, H4 = G.shape
    if head_dy is not None:
        dy8, hw = head_dy
        dHs = torch.bmm(dy8[:, :, :hw.shape[2]].to(hw.dtype), hw.transpose(1, 2)).view(N, T, E, -1)
    H = H4 // 4
    wh = img if wh is None else wh
    dh_rec = None
    dc = torch.zeros(N, E, H, dtype=G.dtype, device=G.device)
    for t in range(T - 1, -1, -1):
        dc_prev, dhd = torch.empty_like(dc), torch.empty_like(dc)
        bptt_step(G[:, t], Call[:, t], Call[:, t + 1], done[t], dHs[:, t], dh_rec, dc, (None, wh, None), dZ[:, t], dc_prev, dhd, True)
        dc, dh_rec = dc_prev, dhd
    db = dZ.reshape(N, T * E, H4).sum(dim=1) if want_db else None
    return db, (dh_rec if want_state_grad else None), (dc if want_state_grad else None)


def nbr_onehot(action, nbr_idx, n_a, out=None):
    """one_hot(boolean_mask(action, mask_i)) -> [N,E,m_max*A] (policies.py:66-68, 305)."""
    E, N = action.shape
    m = nbr_idx.shape[1]
    y = torch.zeros(N, E, m * n_a, dtype=torch.float32, device=action.device)
    for i, js in enumerate(neighbor_lists(nbr_idx)):
        for k, j in enumerate(js):
            y[i, torch.arange(E), k * n_a + action[:, j].long()] = 1.0
    if out is not None:
        out.copy_(y)
        return out
    return y


def lstm_cell(z, bias, c_prev, done):
    """agents/utils.py:102-113: gate order i,f,o,u; state masked by (1-done)."""
    H = z.shape[-1] // 4
    zb = z + bias.unsqueeze(1)
    i, f, o, u = zb.split(H, dim=-1)
    i, f, o, u = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(u)
    keep = (1.0 - done).view(1, -1, 1)
    c = f * (c_prev * keep) + i * u
    h = o * torch.tanh(c)
    return h, c


FUSED_H = 64


XSIDE_MAX_K = 256


def xside_supported(kx, n_h):
    return n_h == FUSED_H and kx % 32 == 0 and 0 <= kx <= XSIDE_MAX_K


MSG_GATHER_RELU, MSG_MEAN_ADD, MSG_DIAL = 1, 2, 3


def msg_supported(kind, m_max, n_h):
    return n_h == FUSED_H and m_max <= 8 and (n_h if kind == MSG_MEAN_ADD else n_h * m_max) <= 128


def ob_encoder_supported(n_feat, n_obs, n_h):
    return n_feat % 4 == 0 and n_obs <= 64


def lstm_ob_wimage(w_ob, pad, out=None):
    return torch.zeros(w_ob.shape[0], 1) if out is None else out


def step_sync_words(N, E, device):
    return torch.zeros(16, dtype=torch.int32)


def step_handoff_supported(N, E, device, K=128):
    """The product's one-launch policy + value step of coupled nets is a kernel-side arrangement; the restatement runs the
    two steps one after the other either way.  NMARL_INKERNEL_HANDOFF=0 selects the host code of the two-launch path."""
    import os
    return os.environ.get('NMARL_INKERNEL_HANDOFF', '1') != '0'


def lstm_msg_wimage(w_msg, out=None):
    return torch.zeros(w_msg.shape[0], 1) if out is None else out


def lstm_wimage(wx, wh, out=None):
    """The product's chunked LDS image of [wx; wh] is a kernel-side layout; the restatement multiplies by wx / wh
    directly, so the 'image' is just a token."""
    return torch.zeros(wh.shape[0], 1) if out is None else out


def lstm_step_fused(h, wh, bias, zadd1, zadd2, c_prev, done, gates, c_out, h_out, xs=None):
    """agents/utils.py:102-113 with z = zadd1 (+ zadd2) + (h*(1-done)) @ wh; with xs = (x, wx, image) the x-side
    product x @ wx is part of the step (z = x @ wx + (h*(1-done)) @ wh (+ zadd1) (+ zadd2))."""
    keep = (1.0 - done).view(1, -1, 1)
    z = torch.bmm(h * keep, wh)
    if xs is not None:
        x = xs[0]
        if len(xs) > 3 and xs[3] is not None:          # [x | x2]: the last columns of the LSTM input from a second tensor
            x = xs[3] if x is None else torch.cat([x, xs[3]], dim=-1)
        if len(xs) > 4 and xs[4] is not None:          # [x | message term] from the neighbours' un-masked h (quirk Q3)
            m = xs[4]
            # the product's one-launch lock-steps hand the value re-step's message term to the next lock-step's policy step
            # (nmarl_msg_t.carry_in / carry_out / mean_next): `_carry_role` says which of the two steps this call restates
            role = m.get('_carry_role')
            cin = m.get('carry_in') if role == 'policy' else None
            if m['kind'] == 1:                         # lstm_comm: relu([h_j] W_msg + b)   (agents/utils.py:182-199)
                t = torch.relu(torch.bmm(nbr_gather(h, m['nbr_idx']), m['w_msg']) + m['b_msg'].unsqueeze(1)) if cin is None else cin.clone()
                if role == 'value' and m.get('carry_out') is not None:
                    m['carry_out'].copy_(t)
            elif m['kind'] == 3:                       # lstm_dial: relu([msg_j] W_msg + b) + enc, msg_j the senders' vectors (agents/utils.py:560-580)
                t = torch.relu(torch.bmm(nbr_gather(m['src'], m['nbr_idx']), m['w_msg']) + m['b_msg'].unsqueeze(1))
                if m.get('out2') is not None:
                    m['out2'].copy_(t)
                t = t + m['enc']
            else:                                      # lstm_ic3: mean_j(h_j) W_msg + b + enc   (agents/utils.py:395-400)
                if m.get('ob') is not None:            # enc = tanh([x_i | x_nbr] W_ob + b_ob) (agents/utils.py:395-399) computed here, kept in m['enc']
                    ob = m['ob']
                    idx = ob['nbr'].long()
                    g = ob['x'][:, idx.clamp(min=0), :] * (idx >= 0).to(ob['x'].dtype).unsqueeze(-1)        # [E,N,slots,F]
                    g = g.reshape(g.shape[0], g.shape[1], -1).transpose(0, 1).to(ob['w'].dtype)             # [N,E,n_obs]
                    m['enc'].copy_(torch.tanh(torch.bmm(g, ob['w']) + ob['b'].unsqueeze(1)))
                if cin is None:
                    mm_ = nbr_mean(h, m['nbr_idx'])
                    if m.get('mean_out') is not None:
                        m['mean_out'].copy_(mm_)
                    if role == 'value' and m.get('mean_next') is not None:
                        m['mean_next'].copy_(mm_)
                    t0 = torch.bmm(mm_, m['w_msg']) + m['b_msg'].unsqueeze(1)
                else:                                  # (the previous launch wrote this lock-step's mean rows: mean_next)
                    t0 = cin.clone()
                if role == 'value' and m.get('carry_out') is not None:
                    m['carry_out'].copy_(t0)
                t = t0 + m['enc']
            if m.get('out') is not None:
                m['out'].copy_(t)
            x = t if x is None else torch.cat([x, t], dim=-1)
        if x is not None:
            z = z + torch.bmm(x, xs[1])
    if zadd1 is not None:
        z = z + zadd1
    if zadd2 is not None:
        z = z + zadd2
    hn, cn = lstm_cell(z, bias, c_prev, done)
    if gates is not None:
        H = h.shape[-1]
        zb = z + bias.unsqueeze(1)
        gates.copy_(torch.cat([torch.sigmoid(zb[..., :3 * H]), torch.tanh(zb[..., 3 * H:])], dim=-1))
    c_out.copy_(cn)
    h_out.copy_(hn)
    return h_out, c_out


def lstm_step_policy(h, wh, bias, zadd1, zadd2, c_prev, done, c_out, h_out, pi_w, pi_b, pi_out, act_out, mode,
                     u=None, seed=0, env_id_base=0, step=0, step_dev=None, xs=None, gates=None):
    """forward('p') (policies.py:119-123, 50-57) + the action draw (utils.py:135-141) after one LSTM step."""
    lstm_step_fused(h, wh, bias, zadd1, zadd2, c_prev, done, gates, c_out, h_out, xs=xs)
    nxt = xs[4].get('next') if xs is not None and len(xs) > 4 and xs[4] is not None else None
    if nxt is not None:                                # lstm_dial: the sender layer on the new h (agents/utils.py:566-569)
        nxt['out'].copy_(torch.relu(torch.bmm(h_out, nxt['w']) + nxt['b'].unsqueeze(1)))
    pi_out.copy_(torch.softmax(torch.bmm(h_out, pi_w) + pi_b.unsqueeze(1), dim=-1))
    sample_actions(pi_out, act_out, mode, u=u, seed=seed, env_id_base=env_id_base, step=step, step_dev=step_dev)
    return pi_out, act_out


def lstm_step_value(h, wh, bias, zadd1, zadd2, c_prev, done, c_out, h_out, v_w, v_b, action, nbr_idx, n_a, v_out, xs=None):
    """forward('v') (policies.py:124-133, 59-77): v = [h', one_hot(neighbour actions)] @ w + b."""
    lstm_step_fused(h, wh, bias, zadd1, zadd2, c_prev, done, None, c_out, h_out, xs=xs)
    na = nbr_onehot(action, nbr_idx, n_a)
    v_out.copy_((torch.bmm(torch.cat([h_out, na], dim=-1), v_w) + v_b.unsqueeze(1)).squeeze(-1))
    return v_out


BIAS_NONE, BIAS_RELU, BIAS_TANH = 0, 1, 2


FC_MAX_F, FC_J = 64, 64


def _act(t, act):
    return torch.relu(t) if act == BIAS_RELU else torch.tanh(t) if act == BIAS_TANH else t


def fc_supported(x, w):
    return x.shape[2] <= FC_MAX_F and w.shape[2] == FC_J


def fc_fwd(x, w, b, act, out=None):
    """fc (agents/utils.py:65-73): act(x @ w + b) per agent."""
    y = _act(torch.bmm(x, w) + b.unsqueeze(1), act)
    if out is not None:
        out.copy_(y)
        return out
    return y


def onehot_argmax_add_(y, p, scale=None):
    """agents/utils.py:577-579: y += one_hot(argmax(p), n_h) (per-agent factor `scale`: lstm_dial_hetero, :676-688)."""
    oh = torch.nn.functional.one_hot(torch.argmax(p, dim=-1), y.shape[-1]).to(y.dtype)
    if scale is not None:
        oh = oh * scale.view(-1, 1, 1)
    return y.add_(oh)


def fc_fwd_multi(parts, act, out=None):
    """tf.concat of fc layers, a layer's input optionally gathered over the neighbour table (models.py:171-179)."""
    ys = [fc_fwd(x if nbr_idx is None else nbr_gather(x, nbr_idx), w, b, act) for x, w, b, nbr_idx in parts]
    y = torch.cat(ys, dim=-1)
    if out is not None:
        out.copy_(y)
        return out
    return y


def fc_bwd(x, y, dy, act, nbr_idx=None):
    """Gradient of fc w.r.t. (w, b), the activation derivative taken from the layer output y."""
    if nbr_idx is not None:
        x = nbr_gather(x, nbr_idx)
    g = dy * (y > 0).to(dy.dtype) if act == BIAS_RELU else dy * (1.0 - y * y) if act == BIAS_TANH else dy
    return torch.bmm(x.transpose(1, 2), g), g.sum(1)


def fc_concat(parts, act, saved=None, bits=None):
    """tf.concat of per-input fc layers (policies.py:176-181, agents/utils.py:186-199), plain autograd."""
    parts = [tuple(pt) + (None,) * (4 - len(pt)) for pt in parts]
    ys = [_act(torch.baddbmm(b.unsqueeze(1), x if idx is None else nbr_gather(x, idx), w), act) for x, w, b, idx in parts]
    return ys[0] if len(ys) == 1 else torch.cat(ys, dim=-1)


def thin_linear_bwd(h, dy, w, dy2=None):
    """Gradient of y = h @ w + b (the heads, policies.py:50-77); dy2: separate gradient of the last column."""
    if dy2 is not None:
        dy = torch.cat([dy, dy2.unsqueeze(-1)], dim=-1)
    return torch.bmm(dy, w.transpose(1, 2)), torch.bmm(h.transpose(1, 2), dy), dy.sum(1)


def thin_linear(h, w, b):
    return torch.baddbmm(b.unsqueeze(1), h, w)


def nbr_action_value(action, nbr_idx, w_a, n_a, out=None, accumulate=False):
    """one_hot(neighbour actions) @ w_a (policies.py:66-72)."""
    N = action.shape[1]
    va = torch.bmm(nbr_onehot(action, nbr_idx, n_a).to(w_a.dtype), w_a.reshape(N, -1, 1)).squeeze(-1)
    if out is None:
        return va
    out.copy_(out + va if accumulate else va)
    return out


def step_enc_supported(n_feat, n_a, m_max, n_fc, n_h, N):
    """(the product's csrc/lstm_mfma.hip ENC pre-phase: the CACC input layout of IA2C-FP)"""
    return n_feat == 5 and n_a == 4 and m_max == 2 and n_fc == 64 and n_h == 64 and N <= 32


def step_enc1_supported(n_feat, m_max, n_fc, n_h, N):
    """(ENC 2: the observation encoder alone -- IA2C with two neighbour slots, ConseNet with none)"""
    return n_feat == 5 and m_max in (0, 2) and n_fc == 64 and n_h == 64 and N <= 32


def step_enc_spec(ob, fp, w_ob, b_ob, w_fp, b_fp, nbrs, out=None, env=None, bits=None):
    assert env is None, 'the in-launch env step exists on the device only (tests compare it with the env kernel there)'
    return dict(ob=ob, fp=fp, w_ob=w_ob, b_ob=b_ob, w_fp=w_fp, b_fp=b_fp, nbrs=nbrs, out=out, bits=bits)


def relu_bits_pack(S):
    """Sign image of S [..., 128] as the product's lock-step kernel writes it (include/nmarl.h nmarl_step_enc_t.relu_bits):
    [..., 4] int32, bit 4 t + i of word q <=> S[..., 16 t + 4 q + i] > 0."""
    pos = (S > 0).reshape(*S.shape[:-1], 8, 4, 4).to(torch.int64)
    sh = 4 * torch.arange(8).view(8, 1, 1) + torch.arange(4).view(1, 1, 4)
    w = (pos << sh.to(S.device)).sum(dim=(-3, -1))
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def step_enc_forward(d):
    """FPPolicy's input encoders (policies.py:176-181) from the env's COMPACT observation ob [E,N,5] and the previous-step
    policies fp [N,E,4]: s_i = [relu([x_i | x_nbr] W_ob + b_ob) | relu([pi_nbr] W_fp + b_fp)], neighbours in ascending index,
    left packed, absent slots zero -> [N,E,128]; also written to d['out'] when given."""
    ob, fp, nbrs = d['ob'], d['fp'], d['nbrs']
    E, N, F = ob.shape
    single = d.get('w_fp') is None            # the observation encoder alone (IA2C: [own | 2 neighbours]; ConseNet: own features only)
    own_only = single and d['w_ob'].shape[1] == F
    A = 0 if single else fp.shape[2]
    rows = []
    for i in range(N):
        nb = list(nbrs[i]) + [-1] * (2 - len(nbrs[i]))
        xo = ob[:, i] if own_only else \
            torch.cat([ob[:, i]] + [ob[:, j] if j >= 0 else torch.zeros(E, F, dtype=ob.dtype, device=ob.device) for j in nb], dim=1)
        hx = torch.relu(xo @ d['w_ob'][i] + d['b_ob'][i])
        if single:
            rows.append(hx)
            continue
        xf = torch.cat([fp[j] if j >= 0 else torch.zeros(E, A, dtype=fp.dtype, device=fp.device) for j in nb], dim=1)
        rows.append(torch.cat([hx, torch.relu(xf @ d['w_fp'][i] + d['b_fp'][i])], dim=1))
    S = torch.stack(rows, dim=0)
    if d.get('out') is not None:
        d['out'].copy_(S)
    if d.get('bits') is not None:
        d['bits'].copy_(relu_bits_pack(S))
    return S


def lstm_step_policy_value(h, wh, bias, zadd1, zadd2, c, done, pi_w, pi_b, pi_out, act_out, v_w, v_b, nbr_idx, n_a,
                           v_out, mode, u=None, seed=0, env_id_base=0, step=0, step_dev=None, xs=None, h_out=None,
                           c_out=None, gates=None, defer_action_term=False):
    """Trainer._get_policy + _get_value of one lock-step (utils.py:129-149): forward('p') advances the state, forward('v')
    re-steps a COPY of it (policies.py:119-133, quirk Q1)."""
    h_out, c_out = (h if h_out is None else h_out), (c if c_out is None else c_out)
    if xs is not None and isinstance(xs[0], dict):       # the input encoders run inside the product's launch
        with torch.no_grad():
            xs = (step_enc_forward(xs[0]),) + tuple(xs[1:])
    if xs is not None and len(xs) > 4 and xs[4] is not None and xs[4].get('enc_spec') is not None:
        # lstm_comm's one-launch lock-step: the input encoders write [hx | hp] into the x slot before anything reads it
        with torch.no_grad():
            step_enc_forward(dict(xs[4]['enc_spec'], out=xs[0]))
        xs = tuple(xs[:4]) + ({k: v for k, v in xs[4].items() if k != 'enc_spec'},)
    xs_p = xs
    if xs is not None and len(xs) > 4 and xs[4] is not None:
        # coupled net: only the POLICY step's message term is kept (`out`); the re-step's comes from the new h of all agents
        xs_p = tuple(xs[:4]) + (dict(xs[4], _carry_role='policy'),)
        xs = tuple(xs[:4]) + (dict({k: v for k, v in xs[4].items() if k not in ('out', 'mean_out')}, _carry_role='value'),)
    if gates is not None:
        xs_g = xs
        if xs_p is not xs:                   # (the gates are the POLICY step's: its message term, without its outputs)
            xs_g = tuple(xs[:4]) + (dict(xs[4], _carry_role='policy'),)
        lstm_step_fused(h, wh, bias, zadd1, zadd2, c, done, gates, torch.empty_like(c), torch.empty_like(h), xs=xs_g)
        if xs_p is not xs:                   # the step below writes (h_out, c_out); the message term needs the OLD h of all agents
            assert h_out.data_ptr() != h.data_ptr()
    lstm_step_policy(h, wh, bias, zadd1, zadd2, c, done, c_out, h_out, pi_w, pi_b, pi_out, act_out, mode, u=u, seed=seed,
                     env_id_base=env_id_base, step=step, step_dev=step_dev, xs=xs_p)
    if defer_action_term:      # only the h part of the critic: v = h'' @ w[:H] + b
        H = h.shape[-1]
        hv, cv = torch.empty_like(h), torch.empty_like(c)
        lstm_step_fused(h_out, wh, bias, zadd1, zadd2, c_out, done, None, cv, hv, xs=xs)
        v_out.copy_((torch.bmm(hv, v_w[:, :H]) + v_b.unsqueeze(1)).squeeze(-1))
    else:
        lstm_step_value(h_out, wh, bias, zadd1, zadd2, c_out, done, torch.empty_like(c), torch.empty_like(h), v_w, v_b,
                        act_out, nbr_idx, n_a, v_out, xs=xs)
    return pi_out, act_out, v_out


def nbr_action_value_bwd(action, nbr_idx, dv, n_a):
    return torch.bmm(nbr_onehot(action, nbr_idx, n_a).to(dv.dtype).transpose(1, 2), dv.unsqueeze(-1)).squeeze(-1)


def heads_supported(h, n_a, nbr_idx):
    return True


def heads(h, pi_w, pi_b, v_w, v_b, action, nbr_idx, n_a):
    """policies.py:50-77, plain autograd: logits = h pi_w + pi_b; v = [h, one_hot(na)] v_w + v_b."""
    H = h.shape[2]
    logits = torch.baddbmm(pi_b.unsqueeze(1), h, pi_w)
    na = nbr_onehot(action, nbr_idx, n_a).to(h.dtype)
    v = torch.baddbmm(v_b.unsqueeze(1), torch.cat([h, na], dim=-1), v_w).squeeze(-1)
    return logits, v


def wgrad(a, g, out=None):
    """a^T g over all rows (the weight gradient of a batched layer)."""
    r = torch.bmm(a.transpose(1, 2), g)
    if out is None:
        return r
    out.copy_(r)
    return out


def linear(x, w):
    return torch.bmm(x, w)


def bias_act_(x, bias, act, out=None):
    """fc's bias + activation (agents/utils.py:65-73), in place or into `out`."""
    y = x + bias.unsqueeze(1)
    y = torch.relu(y) if act == BIAS_RELU else torch.tanh(y) if act == BIAS_TANH else y
    (x if out is None else out).copy_(y)
    return x if out is None else out


def lstm_cell_infer(z, bias, c_prev, done, c_out, h_out, z2=None):
    h, c = lstm_cell(z if z2 is None else z + z2, bias, c_prev, done)
    c_out.copy_(c)
    h_out.copy_(h)
    return h_out, c_out


def lstm_sequence(pre, wh, b, h0, c0, done, masked_steps=None):
    """agents/utils.py:102-113 over T steps for all agents: plain autograd loop."""
    N, T, E, H4 = pre.shape
    h, c = h0, c0
    hs = []
    for t in range(T):
        keep = (1.0 - done[t]).view(1, E, 1)
        z = pre[:, t] + torch.bmm(h * keep, wh)
        h, c = lstm_cell(z, b, c, done[t])
        hs.append(h)
    return torch.stack(hs, dim=1)


def lstm_sequence_x(s, wx, wh, b, h0, c0, done, masked_steps, img):
    """lstm_sequence with the x-side product inside the step: s [N,T,E,KX], wx [N,KX,4H]."""
    N, T, E, KX = s.shape
    h, c = h0, c0
    hs = []
    for t in range(T):
        keep = (1.0 - done[t]).view(1, E, 1)
        z = torch.bmm(s[:, t], wx) + torch.bmm(h * keep, wh)
        h, c = lstm_cell(z, b, c, done[t])
        hs.append(h)
    return torch.stack(hs, dim=1)


def lstm_sequence_saved(s, wx, wh, b, G, Hall, Call, done, masked_steps, s_ext=None):
    """The restatement has no saved-activation shortcut: it recomputes the sequence from Hall[:, 0] / Call[:, 0] (what
    the product's rollout saved must equal this; tests compare the two)."""
    return lstm_sequence_x(s, wx, wh, b, Hall[:, 0], Call[:, 0], done, masked_steps, None)


SAMPLE_UNIFORM, SAMPLE_PHILOX, SAMPLE_ARGMAX = 0, 1, 2


def sample_actions(pi, out, mode, u=None, seed=0, env_id_base=0, step=0, step_dev=None):
    """np.random.choice == searchsorted(cumsum(p)/sum, u, 'right') (utils.py:138); argmax (utils.py:140)."""
    from oracle import philox
    N, E, A = pi.shape
    if step_dev is not None:
        step = int(step) + int(step_dev.item())       # slot offset + device batch base
    p = pi.detach().cpu().numpy().astype(np.float64).transpose(1, 0, 2)       # [E,N,A]
    if mode == SAMPLE_ARGMAX:
        a = p.argmax(-1)
    else:
        if mode == SAMPLE_UNIFORM:
            uu = u.detach().cpu().numpy().astype(np.float64).reshape(E, N)
        else:
            uu = philox.action_uniform(seed, env_id_base + np.arange(E), N, step).astype(np.float64)
        cdf = np.cumsum(p, axis=-1)
        cdf = cdf / cdf[..., -1:]
        a = (cdf <= uu[..., None]).sum(-1)
        a = np.minimum(a, A - 1)
    out.copy_(torch.from_numpy(a.astype(np.uint8)))
    return out


def a2c_loss_supported(n_a):
    return True


def a2c_loss(logits, v, action, adv, R, v_coef, e_coef):
    """Policy.prepare_loss (policies.py:20-30; 232-255), per agent, batch mean over rows; plain autograd."""
    pi = torch.softmax(logits, dim=-1)
    acts = action.t().long().unsqueeze(-1)                               # [N, rows, 1]
    log_pi = torch.log(torch.clamp(pi, 1e-10, 1.0))
    entropy = -(pi * log_pi).sum(-1)
    logp_a = log_pi.gather(-1, acts).squeeze(-1)
    policy_loss = -(logp_a * adv).mean(-1)
    value_loss = (R - v).pow(2).mean(-1) * 0.5 * v_coef
    entropy_loss = -entropy.mean(-1) * e_coef
    terms = torch.stack([policy_loss, value_loss, entropy_loss], dim=1)
    return terms.sum(dim=1), terms.detach()


def nstep_return(r, v, done_post, R_end, gamma, alpha, dist=None, R_out=None, adv_out=None):
    """OnPolicyBuffer._add_R_Adv / _add_s_R_Adv (agents/utils.py:763-775, 800-816) per (agent, replica)."""
    T, N, E = v.shape
    r64, v64 = r.double().cpu(), v.double().cpu()
    keep = 1.0 - done_post.double().cpu()
    R = R_end.double().cpu().clone()                      # [N,E]
    Rs = torch.zeros(N, T, E, dtype=torch.float64)
    if alpha >= 0:
        w = torch.pow(torch.tensor(float(alpha), dtype=torch.float64), dist.double().cpu())   # [N,N]
        w = torch.where(dist.cpu() < 0, torch.zeros_like(w), w)     # unreachable pairs (-1) match no hop count t
    for t in range(T - 1, -1, -1):
        if alpha < 0:
            R = r64[t].unsqueeze(0) + gamma * R * keep[t].unsqueeze(0)
        else:
            R = gamma * R * keep[t].unsqueeze(0) + w @ r64[t].t()            # [N,N] @ [N,E]
        Rs[:, t] = R
    adv = Rs - v64.permute(1, 0, 2)
    Rs32, adv32 = Rs.float().to(v.device), adv.float().to(v.device)
    if R_out is not None:
        R_out.copy_(Rs32)
        adv_out.copy_(adv32)
        return R_out, adv_out
    return Rs32, adv32


def rmsprop_tf_clip(w, g, ms, scratch, lr, rho, eps, max_norm, grad_scale=1.0, norm_out=None, lr_dev=None, guard=False):
    """tf.clip_by_global_norm + ApplyRMSProp (policies.py:32-39), per row (= optimiser) of [G,P]."""
    with torch.no_grad():
        gs = g * grad_scale
        norm = gs.pow(2).sum(dim=1, keepdim=True).sqrt()
        if max_norm > 0:
            gs = gs * (max_norm * torch.minimum(1.0 / norm, torch.full_like(norm, 1.0 / max_norm)))
        ms.add_((gs * gs - ms) * (1.0 - rho))
        w.sub_(lr * gs / torch.sqrt(ms + eps))
        if norm_out is not None:
            norm_out[:norm.shape[0]].copy_(norm.view(-1))


# ---- the coupled nets' reverse recurrence in one op (csrc/lstm_bptt.hip: lstm_bptt_coupled_kernel)
COUPLED_NC, COUPLED_IC3 = 1, 2


def lstm_bptt_msg_wimage(w_msg, out=None):
    return w_msg                   # the restatement multiplies by the weight itself


def reverse_neighbor_table(nbr_idx, kind):
    return dict(nbr_idx=nbr_idx, r_max=1, r_row=2, symmetric=True)


def dial_adjoint_supported(m_max, H, rev):
    return H == FUSED_H and m_max <= 4 and rev is not None


def dial_adjoint_images(w_msg, mfc_w):
    return None                    # the restatement multiplies by the weights themselves


def dial_adjoint_bias_parts(N, E, device):
    return torch.zeros(N, 1, FUSED_H), torch.zeros(N, 1, FUSED_H)


def dial_msg_adjoint(ds, hm, msg, dhd, w_msg, mfc_w, nbr_idx, imgs, rev, d1, d2, dh, bias_parts=None):
    """Backward of lstm_dial's message path for one step (agents/utils.py:560-580 read backwards): relu mask of the
    receiver layer, d1 @ w_msg^T scattered back to the senders (adjoint of the neighbour gather), relu mask of the sender
    layer, + the recurrent part: dh = dhd + d2 @ mfc_w^T."""
    H = ds.shape[-1]
    d1.copy_(ds * (hm > 0).to(ds.dtype))
    dmsg = nbr_gather_bwd(torch.bmm(d1, w_msg.transpose(1, 2)), nbr_idx, H)
    d2.copy_(dmsg * (msg > 0).to(ds.dtype))
    dh.copy_(dhd + torch.bmm(d2, mfc_w.transpose(1, 2)))
    if bias_parts is not None:
        bias_parts[0][:, 0] += d1.sum(dim=1).to(bias_parts[0].dtype)
        bias_parts[1][:, 0] += d2.sum(dim=1).to(bias_parts[1].dtype)
    return dh


def bptt_coupled_supported(kind, m_max, H, rev=None):
    return ((kind == COUPLED_NC and m_max <= 2) or kind == COUPLED_IC3) and not (
        rev is not None and kind == COUPLED_NC and rev['r_max'] > 2)


def bptt_coupled(kind, rev, m_max, G, Call, done, dHs, ws, wm, mask, dZ, D1, mode=0, head_dy=None):
    """Restatement of nmarl_lstm_bptt_coupled: per reverse step cell backward, [dx | dh] = dz @ [wxm; wh]^T, D1 = dx (relu
    mask for lstm_comm), message adjoint through the neighbour table; ws = (wxm, wh, .), wm = (w_msg, .).
    head_dy = (dy8 [N,T*E,8], hw [N,64,O]): the heads' dL/dh = dy hw^T (the product's kernel forms it itself)."""
    if head_dy is not None:
        dy8, hw = head_dy
        dHs = torch.bmm(dy8[:, :, :hw.shape[2]].to(hw.dtype), hw.transpose(1, 2)).view(G.shape[0], G.shape[1], G.shape[2], -1)
    wxm, wh, _ = ws
    w_msg = wm[0]
    nbr_idx = rev['nbr_idx']
    N, T, E, H4 = G.shape
    H = H4 // 4
    dh_rec = None
    dc = torch.zeros(N, E, H, dtype=G.dtype)
    for t in range(T - 1, -1, -1):
        dz_t, dc_prev = torch.empty(N, E, H4, dtype=G.dtype), torch.empty(N, E, H, dtype=G.dtype)
        cell_bwd(G[:, t], Call[:, t], Call[:, t + 1], done[t], dHs[:, t], dc, dz_t, dc_prev, dh2=dh_rec)
        dc = dc_prev
        dZ[:, t].copy_(dz_t)
        dhd = torch.bmm(dz_t, wh.transpose(1, 2)) * (1.0 - done[t]).view(1, -1, 1)
        dx = torch.bmm(dz_t, wxm.transpose(1, 2))
        if kind == COUPLED_NC:
            dx = dx * (mask[:, t] > 0)
        D1[:, t].copy_(dx)
        m_t = torch.bmm(dx, w_msg.transpose(1, 2))
        dh_rec = (nbr_gather_bwd(m_t, nbr_idx, H) if kind == COUPLED_NC else nbr_mean_bwd(m_t, nbr_idx)) + dhd
    return dZ.sum(dim=(1, 2)), D1.sum(dim=(1, 2))


def batch_epilogue(g, done, ep_sum, ep_sq, ep_len, fin, T_env, h_fw, c_fw, h_bw, c_bw, fp_T, fp_0, fp_uniform, x_T, x_0, done_pre,
                   skip_if=None):
    """Restatement of nmarl_batch_epilogue (csrc/a2c.hip): the elementwise host code the batched loop used to run."""
    if skip_if is not None and int(skip_if.reshape(-1)[0]) != 0:
        return
    T = g.shape[0]
    gd = g.double()
    ep_sum += gd.sum(0)
    ep_sq += (gd * gd).sum(0)
    ep_len += T
    dmf = done.bool().double()
    mean = ep_sum / ep_len
    std = (ep_sq / ep_len - mean * mean).clamp_min(0).sqrt()
    coll = (ep_len < T_env).double() * dmf
    fin += torch.stack([dmf.sum(), (mean * dmf).sum(), (std * dmf).sum(), coll.sum()])
    keep = 1.0 - dmf
    ep_sum *= keep
    ep_sq *= keep
    ep_len *= keep
    k32 = keep.to(h_fw.dtype).view(1, -1, 1)
    for s_ in (h_fw, c_fw):
        s_.mul_(k32)
    h_bw.copy_(h_fw)
    c_bw.copy_(c_fw)
    fp_0.copy_(fp_T * k32 + (1.0 - k32) * fp_uniform.reshape(fp_T.shape[0], 1, -1))
    x_0.copy_(x_T)
    done_pre.copy_(done.to(done_pre.dtype))
