"""Fused n_step recurrences with CROSS-AGENT coupling (NeurComm, CommNet, DIAL) for the update: manual BPTT as one
torch.autograd.Function instead of ~15 autograd nodes per step.

Reference recurrences (agents/utils.py): lstm_comm 182-208, lstm_ic3 385-408, lstm_dial 561-593.  Per step t, with
h = h_{t-1} (un-masked for the messages, quirk Q3), `enc_t` the h-independent part computed for all T beforehand:

  nc    hm = relu(gather(h) W_msg + b_msg);                     z = enc_t + hm Wx[2H:3H] + (h keep) Wh
  ic3   s  = enc_t + mean_nbr(h) W_msg + b_msg;                 z = s Wx + (h keep) Wh
  dial  hm = relu(gather(relu(h W_mfc + b_mfc)) W_msg + b_msg); z = (enc_t + hm) Wx + (h keep) Wh

followed by the cell (fused MFMA step kernel when H = 64).  Backward: one reverse loop of {cell_bwd, ONE dgrad GEMM
against the adjacent [Wx-part; Wh] rows of the flat parameter buffer, message adjoints}, then every weight gradient as
a SINGLE GEMM over all T*E rows and every bias gradient as a single reduction.
"""
import torch

from .. import ops

F32 = torch.float32


def _stack_rows(w_top, w_bot):
    """[w_top; w_bot] as one [N, rows, 4H] view when the two tensors are adjacent rows of the flat buffer."""
    if (w_top.stride() == w_bot.stride() and w_top.shape[2] == w_bot.shape[2] and w_top.stride(2) == 1 and
            w_top.stride(1) == w_top.shape[2] and
            w_top.data_ptr() + w_top.shape[1] * w_top.shape[2] * 4 == w_bot.data_ptr()):
        return torch.as_strided(w_top, (w_top.shape[0], w_top.shape[1] + w_bot.shape[1], w_top.shape[2]), w_top.stride())
    return None


class CoupledSequence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, nbr_idx, masked_steps, enc, h0, c0, done, wx, wh, b, w_msg, b_msg, mfc_w, mfc_b):
        N, T, E, _ = enc.shape
        H = h0.shape[-1]
        dev = enc.device
        G = torch.empty(N, T, E, 4 * H, dtype=F32, device=dev)
        Hall = torch.empty(N, T + 1, E, H, dtype=F32, device=dev)
        Call = torch.empty(N, T + 1, E, H, dtype=F32, device=dev)
        Hall[:, 0].copy_(h0)
        Call[:, 0].copy_(c0)
        A1 = torch.empty(N, T, E, H, dtype=F32, device=dev)          # nc/dial: hm (post-relu); ic3: s
        A2 = torch.empty(N, T, E, H, dtype=F32, device=dev) if kind == 'dial' else None   # dial: msg (post-relu)
        S = torch.empty(N, T, E, H, dtype=F32, device=dev) if kind == 'dial' else None    # dial: enc + hm
        masked = set(range(T)) if masked_steps is None else set(masked_steps)
        fused = H == ops.FUSED_H
        keep = 1.0 - done
        for t in range(T):
            hp = Hall[:, t].contiguous()
            if kind == 'nc':
                ops.bias_act_(torch.bmm(ops.nbr_gather(hp, nbr_idx), w_msg), b_msg, ops.BIAS_RELU, out=A1[:, t])
                z1, z2 = enc[:, t], torch.bmm(A1[:, t], wx)
            elif kind == 'ic3':
                s = ops.bias_act_(torch.bmm(ops.nbr_mean(hp, nbr_idx), w_msg), b_msg, ops.BIAS_NONE)
                torch.add(s, enc[:, t], out=A1[:, t])
                z1, z2 = torch.bmm(A1[:, t], wx), None
            else:
                ops.bias_act_(torch.bmm(hp, mfc_w), mfc_b, ops.BIAS_RELU, out=A2[:, t])
                ops.bias_act_(torch.bmm(ops.nbr_gather(A2[:, t].contiguous(), nbr_idx), w_msg), b_msg, ops.BIAS_RELU,
                              out=A1[:, t])
                torch.add(A1[:, t], enc[:, t], out=S[:, t])
                z1, z2 = torch.bmm(S[:, t], wx), None
            if fused:
                ops.lstm_step_fused(Hall[:, t], wh, b, z1, z2, Call[:, t], done[t], G[:, t], Call[:, t + 1], Hall[:, t + 1])
            else:
                hk = hp * keep[t].view(1, E, 1) if t in masked else hp
                z = torch.bmm(hk, wh) if z2 is None else torch.baddbmm(z2, hk, wh)
                ops.cell_fwd(z, b, Call[:, t], done[t], G[:, t], Call[:, t + 1], Hall[:, t + 1], z2=z1)
        ctx.save_for_backward(G, Hall, Call, A1, A2 if A2 is not None else G.new_empty(0), S if S is not None else G.new_empty(0),
                              done, wx, wh, w_msg, mfc_w if mfc_w is not None else G.new_empty(0), nbr_idx)
        ctx.kind, ctx.masked = kind, masked
        return Hall[:, 1:]

    @staticmethod
    def backward(ctx, dHs):
        G, Hall, Call, A1, A2, S, done, wx, wh, w_msg, mfc_w, nbr_idx = ctx.saved_tensors
        kind, masked = ctx.kind, ctx.masked
        N, T, E, H4 = G.shape
        H = H4 // 4
        dev = G.device
        dHs = dHs.contiguous()
        dZ = torch.empty_like(G)
        D1 = torch.empty(N, T, E, H, dtype=F32, device=dev)   # nc/dial: d(pre-relu of hm); ic3: ds
        D2 = torch.empty(N, T, E, H, dtype=F32, device=dev) if kind == 'dial' else None   # dial: d(pre-relu of msg)
        DS = torch.empty(N, T, E, H, dtype=F32, device=dev) if kind == 'dial' else None   # dial: ds
        keep = 1.0 - done
        w2 = _stack_rows(wx, wh)                              # [N, 2H, 4H]: one dgrad GEMM gives [d(wx input) | d(h keep)]
        w2_t = None if w2 is None else w2.transpose(1, 2)
        wx_t, wh_t, wmsg_t = wx.transpose(1, 2), wh.transpose(1, 2), w_msg.transpose(1, 2)
        dh_rec = None
        dc = torch.zeros(N, E, H, dtype=F32, device=dev)
        dc_next = torch.empty_like(dc)
        for t in range(T - 1, -1, -1):
            ops.cell_bwd(G[:, t], Call[:, t], Call[:, t + 1], done[t], dHs[:, t], dc, dZ[:, t], dc_next, dh2=dh_rec)
            dc, dc_next = dc_next, dc
            if w2_t is not None:
                d2 = torch.bmm(dZ[:, t], w2_t)
                dx, dhd = d2[..., :H], d2[..., H:]
            else:
                dx, dhd = torch.bmm(dZ[:, t], wx_t), torch.bmm(dZ[:, t], wh_t)
            if t in masked:
                dhd = dhd * keep[t].view(1, E, 1)
            if kind == 'nc':
                torch.mul(dx, (A1[:, t] > 0), out=D1[:, t])
                dh_msg = ops.nbr_gather_bwd(torch.bmm(D1[:, t], wmsg_t), nbr_idx, H)
            elif kind == 'ic3':
                D1[:, t].copy_(dx)
                dh_msg = ops.nbr_mean_bwd(torch.bmm(D1[:, t], wmsg_t), nbr_idx)
            else:
                DS[:, t].copy_(dx)
                torch.mul(dx, (A1[:, t] > 0), out=D1[:, t])
                dmsg = ops.nbr_gather_bwd(torch.bmm(D1[:, t], wmsg_t), nbr_idx, H)
                torch.mul(dmsg, (A2[:, t] > 0), out=D2[:, t])
                dh_msg = torch.bmm(D2[:, t], mfc_w.transpose(1, 2))
            dh_rec = dh_msg + dhd
        R = T * E
        dZf = dZ.view(N, R, H4)
        Hprev = Hall[:, :T]
        if len(masked) == T:
            Hk = (Hprev * keep.view(1, T, E, 1)).reshape(N, R, H)
        else:
            Hk = Hprev.clone()
            for t in masked:
                Hk[:, t].mul_(keep[t].view(1, E, 1))
            Hk = Hk.view(N, R, H)
        dwh = ops.wgrad(Hk, dZf)
        db = dZf.sum(dim=1)
        Hp = Hprev.reshape(N, R, H)                            # un-masked h_{t-1} of all steps (message inputs)
        dmfc_w = dmfc_b = None
        if kind == 'nc':
            dwx = ops.wgrad(A1.view(N, R, H), dZf)
            D1f = D1.view(N, R, H)
            dwmsg = ops.wgrad(ops.nbr_gather(Hp, nbr_idx), D1f)
            dbmsg = D1f.sum(dim=1)
            denc = dZ
        elif kind == 'ic3':
            dwx = ops.wgrad(A1.view(N, R, H), dZf)
            D1f = D1.view(N, R, H)
            dwmsg = ops.wgrad(ops.nbr_mean(Hp, nbr_idx), D1f)
            dbmsg = D1f.sum(dim=1)
            denc = D1
        else:
            dwx = ops.wgrad(S.view(N, R, H), dZf)
            D1f, D2f = D1.view(N, R, H), D2.view(N, R, H)
            dwmsg = ops.wgrad(ops.nbr_gather(A2.view(N, R, H), nbr_idx), D1f)
            dbmsg = D1f.sum(dim=1)
            dmfc_w = ops.wgrad(Hp, D2f)
            dmfc_b = D2f.sum(dim=1)
            denc = DS
        return None, None, None, denc, dh_rec, dc, None, dwx, dwh, db, dwmsg, dbmsg, dmfc_w, dmfc_b


class CoupledSequenceSaved(torch.autograd.Function):
    """The coupled recurrences of the update WITHOUT a forward pass: the rollout's policy steps (x-side step kernel,
    agents/policies.py `_recur_addends`) evaluated exactly this sequence with exactly these weights and saved
    S [N,T,E,KX] (the LSTM inputs: nc [hx | hp | hm], ic3 s, dial enc + hm), the gates G, the state sequences Hall /
    Call and, for dial, the post-relu message terms A1 (hm) / A2 (msg); for ic3 A1 may be the (T + 1)-slab buffer of the
    mean_nbr(h_{t-1}) rows the step kernel kept (the message layer's input: no averaging pass over the h sequence here).  forward = hand out Hall[:, 1:]; backward = the
    manual BPTT of CoupledSequence with the x-side weight as ONE matrix:
      nc    `enc` = [hx | hp] (autograd-connected, = S[..., :2H]); wx = the full [3H,4H]:
            d enc = dZ @ wx[:2H]^T, d wx = S^T dZ (one GEMM over all T*E rows, all three thirds at once)
      ic3 / dial as in CoupledSequence (their `enc` enters the LSTM input additively)."""

    @staticmethod
    def forward(ctx, kind, nbr_idx, masked_steps, enc, done, wx, wh, b, w_msg, b_msg, mfc_w, mfc_b, G, Hall, Call, S, A1, A2,
                s_ext=None):
        T = G.shape[1]
        ctx.s_ext = s_ext
        e = G.new_empty(0)
        ctx.save_for_backward(G, Hall, Call, S, A1 if A1 is not None else e, A2 if A2 is not None else e, done, wx, wh, w_msg,
                              mfc_w if mfc_w is not None else e, nbr_idx)
        ctx.kind = kind
        ctx.masked = set(range(T)) if masked_steps is None else set(masked_steps)
        return Hall[:, 1:]

    @staticmethod
    def backward(ctx, dHs):
        G, Hall, Call, S, A1, A2, done, wx, wh, w_msg, mfc_w, nbr_idx = ctx.saved_tensors
        kind, masked = ctx.kind, ctx.masked
        N, T, E, H4 = G.shape
        H = H4 // 4
        dev = G.device
        R = T * E
        head_dy = ops.take_head_dy(dHs)       # the heads' dL/dh as dy8 (ops.heads_loss): the coupled BPTT kernel expands it itself
        if head_dy is None:
            dHs = dHs.contiguous()
        # (T + 1)-slab operands: with the LSTM inputs handed over as the first T slabs of a (T + 1)-slab buffer (last slab
        # zero), dZ / D1 allocated the same way and the h sequence being (T + 1) slabs anyway, every weight-gradient GEMM
        # reads contiguous [N, (T+1) E, .] operands in place -- no masked copy of the h sequence, no gathered copy
        s_ext = ctx.s_ext
        ext = (s_ext is not None and Hall.is_contiguous() and s_ext.is_contiguous() and
               S.data_ptr() == s_ext.data_ptr() and tuple(s_ext.shape) == (N, T + 1, E, S.shape[-1]) and
               (kind != 'dial' or (tuple(A2.shape) == (N, T + 1, E, H) and A2.is_contiguous())))
        D2e = None
        if ext:
            dZe = torch.empty(N, T + 1, E, H4, dtype=F32, device=dev)
            D1e = torch.empty(N, T + 1, E, H, dtype=F32, device=dev)
            dZe[:, T].zero_()
            D1e[:, T].zero_()
            dZ, D1 = dZe[:, :T], D1e[:, :T]
            if kind == 'dial':
                D2e = torch.empty(N, T + 1, E, H, dtype=F32, device=dev)
                D2e[:, T].zero_()
        else:
            dZ = torch.empty_like(G)
            if kind == 'dial' and tuple(A2.shape) == (N, T + 1, E, H) and A2.is_contiguous():
                # lstm_dial with the senders' vectors in their (T + 1)-slab buffer: D1 the same way (last slab zero), so that the
                # message-weight gradient reads both in place (per run of agents, like lstm_comm) -- no gathered copy of A2
                D1e = torch.empty(N, T + 1, E, H, dtype=F32, device=dev)
                D1e[:, T].zero_()
                D1 = D1e[:, :T]
            else:
                D1e = None
                D1 = torch.empty(N, T, E, H, dtype=F32, device=dev)   # nc/dial: d(pre-relu of hm); ic3: ds
        D2 = (D2e[:, :T] if D2e is not None else torch.empty(N, T, E, H, dtype=F32, device=dev)) if kind == 'dial' else None
        DS = torch.empty(N, T, E, H, dtype=F32, device=dev) if kind == 'dial' else None
        keep = 1.0 - done
        wxm = wx[:, 2 * H:] if kind == 'nc' else wx           # the rows of wx the h-dependent part of the input meets
        hm = S[..., 2 * H:] if kind == 'nc' else A1           # post-relu message term (nc: last third of the input)
        w2 = _stack_rows(wxm, wh)
        w2_t = None if w2 is None else w2.transpose(1, 2)
        wxm_t, wh_t, wmsg_t = wxm.transpose(1, 2), wh.transpose(1, 2), w_msg.transpose(1, 2)
        dh_rec = None
        dc = torch.zeros(N, E, H, dtype=F32, device=dev)
        dc_next = torch.empty_like(dc)
        fused = ops.bptt_supported(H) and wxm.stride(2) == 1 and wxm.stride(1) == H4 and wh.stride(2) == 1 and wh.stride(1) == H4
        ckind = {'nc': ops.COUPLED_NC, 'ic3': ops.COUPLED_IC3}.get(kind)
        rev = None
        db = dbmsg = None
        if fused and ckind is not None and ops.bptt_coupled_supported(ckind, nbr_idx.shape[1], H) and w_msg.stride(2) == 1 and \
                w_msg.stride(1) == H:
            rev = _reverse_table(nbr_idx, ckind)
            if rev is not None and not ops.bptt_coupled_supported(ckind, nbr_idx.shape[1], H, rev=rev):
                rev = None                    # e.g. lstm_comm with more than 2 sources per agent: the step-wise loop below
        if head_dy is not None and rev is None:       # (recurrences without a dy8 form take the tensor it stands for)
            dHs, head_dy = ops.head_dy_to_dh(*head_dy, (N, T, E, H)), None
        if rev is not None:
            # the WHOLE reverse recurrence in one launch: cell backward, [dx | dh] = dz @ [wxm; wh]^T, relu mask, the message
            # adjoint D1 @ w_msg^T handed between the agents' blocks inside the kernel, both bias gradients on the way
            ws = (wxm, wh, ops.lstm_bptt_wimage(wxm, wh))
            wm = (w_msg, ops.lstm_bptt_msg_wimage(w_msg))
            db, dbmsg = ops.bptt_coupled(ckind, rev, nbr_idx.shape[1], G, Call, done, dHs if head_dy is None else None, ws, wm,
                                         hm if kind == 'nc' else None, dZ, D1, head_dy=head_dy)
        elif fused:   # cell backward + [dx | dh] = dz @ [wxm; wh]^T (+ relu mask, + done mask) in one MFMA kernel per step
            ws = (wxm, wh, ops.lstm_bptt_wimage(wxm, wh))
            dhd_buf = torch.empty(N, E, H, dtype=F32, device=dev)
        adj = None
        if fused and kind == 'dial' and w_msg.stride(2) == 1 and w_msg.stride(1) == H:
            # lstm_dial: the whole message adjoint of a step (both relu masks, gather adjoint, both products) in ONE more launch
            rev_d = _reverse_table(nbr_idx, ops.COUPLED_NC)
            if ops.dial_adjoint_supported(nbr_idx.shape[1], H, rev_d):
                adj = (ops.dial_adjoint_images(w_msg, mfc_w), rev_d, torch.empty(N, E, H, dtype=F32, device=dev),
                       ops.dial_adjoint_bias_parts(N, E, dev))
        dbp = ops.bptt_step_db_parts(N, E, H, dev) if (fused and rev is None) else None       # step-wise loop: db on the way
        for t in (range(T - 1, -1, -1) if rev is None else ()):
            if fused:
                ops.bptt_step(G[:, t], Call[:, t], Call[:, t + 1], done[t], dHs[:, t], dh_rec, dc, ws, dZ[:, t], dc_next,
                              dhd_buf, t in masked, dx=DS[:, t] if kind == 'dial' else D1[:, t],
                              mask=hm[:, t] if kind == 'nc' else None, db_part=dbp)
                dc, dc_next = dc_next, dc
                dhd = dhd_buf
                if adj is not None:
                    dh_rec = ops.dial_msg_adjoint(DS[:, t], hm[:, t], A2[:, t], dhd, w_msg, mfc_w, nbr_idx, adj[0], adj[1],
                                                  D1[:, t], D2[:, t], adj[2], bias_parts=adj[3])
                    continue
                if kind == 'dial':
                    torch.mul(DS[:, t], (hm[:, t] > 0), out=D1[:, t])
            else:
                ops.cell_bwd(G[:, t], Call[:, t], Call[:, t + 1], done[t], dHs[:, t], dc, dZ[:, t], dc_next, dh2=dh_rec)
                dc, dc_next = dc_next, dc
                if w2_t is not None:
                    d2 = torch.bmm(dZ[:, t], w2_t)
                    dx, dhd = d2[..., :H], d2[..., H:]
                else:
                    dx, dhd = torch.bmm(dZ[:, t], wxm_t), torch.bmm(dZ[:, t], wh_t)
                if t in masked:
                    dhd = dhd * keep[t].view(1, E, 1)
                if kind == 'ic3':
                    D1[:, t].copy_(dx)
                else:
                    if kind == 'dial':
                        DS[:, t].copy_(dx)
                    torch.mul(dx, (hm[:, t] > 0), out=D1[:, t])
            # dL/dh_{t-1} = message part (adjoint of the neighbour gather / mean) + recurrent part, added in the same pass
            if not dhd.is_contiguous():
                dhd = dhd.contiguous()
            if kind == 'nc':
                dh_rec = ops.nbr_gather_bwd(torch.bmm(D1[:, t], wmsg_t), nbr_idx, H, add=dhd)
            elif kind == 'ic3':
                dh_rec = ops.nbr_mean_bwd(torch.bmm(D1[:, t], wmsg_t), nbr_idx, add=dhd)
            else:
                dmsg = ops.nbr_gather_bwd(torch.bmm(D1[:, t], wmsg_t), nbr_idx, H)
                torch.mul(dmsg, (A2[:, t] > 0), out=D2[:, t])
                dh_rec = torch.baddbmm(dhd, D2[:, t], mfc_w.transpose(1, 2))
        if ext:
            Rx = (T + 1) * E
            Hx, D1x, dZx = Hall.view(N, Rx, H), D1e.view(N, Rx, H), dZe.view(N, Rx, H4)
            # message layer first: its input is the UN-masked h_{t-1} (quirk Q3)
            dmfc_w = dmfc_b = None
            if kind == 'nc':
                dwmsg = _dwmsg_by_runs(Hx, D1x, nbr_idx, H)
            elif kind == 'dial':             # receiver layer on the gathered senders' vectors, sender layer on the un-masked h_{t-1}
                D2x = D2e.view(N, Rx, H)
                dwmsg = _dwmsg_by_runs(A2.view(N, Rx, H), D1x, nbr_idx, H)
                dmfc_w = ops.wgrad(Hx, D2x)
                dmfc_b = D2x.sum(dim=1) if adj is None else adj[3][1].sum(dim=1)
            elif A1.numel() and tuple(A1.shape) == (N, T + 1, E, H) and A1.is_contiguous():
                dwmsg = ops.wgrad(A1.view(N, Rx, H), D1x)       # ic3: the rollout kept mean_nbr(h_{t-1}) (A1 = the (T + 1)-slab MM buffer)
            else:
                dwmsg = ops.wgrad(ops.nbr_mean(Hx, nbr_idx), D1x)
            if db is None:
                db = dZx.sum(dim=1) if dbp is None else dbp.sum(dim=1)
                dbmsg = D1x.sum(dim=1) if adj is None else adj[3][0].sum(dim=1)
            with torch.no_grad():            # h_{t-1} keep_t for the recurrent weight, in the saved buffer itself (the next
                for t in masked:             # rollout rewrites it); steps outside `masked` have done_t = 0 by contract
                    Hall[:, t].mul_(keep[t].view(1, E, 1))
            dwh = ops.wgrad(Hx, dZx)
            dwx = ops.wgrad(s_ext.view(N, Rx, S.shape[-1]), dZx)
            if kind == 'nc':
                denc = None
                if ctx.needs_input_grad[3]:
                    denc = torch.bmm(dZ.reshape(N, R, H4), wx[:, :2 * H].transpose(1, 2)).view(N, T, E, 2 * H)
            else:
                denc = DS if kind == 'dial' else D1
            return (None, None, None, denc, None, dwx, dwh, db, dwmsg, dbmsg, dmfc_w, dmfc_b, None, None, None, None, None, None, None)
        dZf = dZ.view(N, R, H4)
        Hprev = Hall[:, :T]
        if len(masked) == T:
            Hk = (Hprev * keep.view(1, T, E, 1)).reshape(N, R, H)
        else:
            Hk = Hprev.clone()
            for t in masked:
                Hk[:, t].mul_(keep[t].view(1, E, 1))
            Hk = Hk.view(N, R, H)
        dwh = ops.wgrad(Hk, dZf)
        Hp = Hprev.reshape(N, R, H)                            # un-masked h_{t-1} of all steps (message inputs)
        D1f = D1.view(N, R, H) if D1e is None else None        # (lstm_dial with the padded D1: only its (T + 1)-slab view is used)
        if db is None:
            db = dZf.sum(dim=1) if dbp is None else dbp.sum(dim=1)
            if adj is not None:
                dbmsg = adj[3][0].sum(dim=1)                   # lstm_dial: summed inside the adjoint kernel
            else:
                dbmsg = D1f.sum(dim=1) if D1f is not None else D1e.view(N, (T + 1) * E, H).sum(dim=1)
        dwx = ops.wgrad(S.view(N, R, S.shape[-1]), dZf)        # the whole x-side weight in one GEMM
        dmfc_w = dmfc_b = None
        if kind == 'nc':
            dwmsg = ops.wgrad(ops.nbr_gather(Hp, nbr_idx), D1f)
            denc = torch.bmm(dZf, wx[:, :2 * H].transpose(1, 2)).view(N, T, E, 2 * H) if ctx.needs_input_grad[3] else None
        elif kind == 'ic3':
            if A1.numel() and tuple(A1.shape) == (N, T + 1, E, H):
                dwmsg = ops.wgrad(A1[:, :T].reshape(N, R, H), D1f)
            else:
                dwmsg = ops.wgrad(ops.nbr_mean(Hp, nbr_idx), D1f)
            denc = D1
        else:
            D2f = D2.view(N, R, H)
            if D1e is not None:
                dwmsg = _dwmsg_by_runs(A2.view(N, (T + 1) * E, H), D1e.view(N, (T + 1) * E, H), nbr_idx, H)
            else:
                dwmsg = ops.wgrad(ops.nbr_gather(A2[:, :T].reshape(N, R, H), nbr_idx), D1f)
            dmfc_w = ops.wgrad(Hp, D2f)
            dmfc_b = D2f.sum(dim=1) if adj is None else adj[3][1].sum(dim=1)
            denc = DS
        return (None, None, None, denc, None, dwx, dwh, db, dwmsg, dbmsg, dmfc_w, dmfc_b, None, None, None, None, None, None, None)


_rev_tables = {}


def _reverse_table(nbr_idx, ckind):
    """ops.reverse_neighbor_table, built once per (neighbour table, kind): it lives as long as the policy's table."""
    key = (nbr_idx.data_ptr(), nbr_idx.device, ckind, tuple(nbr_idx.shape))
    if key not in _rev_tables:
        _rev_tables[key] = (nbr_idx, ops.reverse_neighbor_table(nbr_idx, ckind))      # keeps nbr_idx alive: the key stays unique
    return _rev_tables[key][1]


_runs = {}


def _dwmsg_by_runs(Hx, D1x, nbr_idx, H):
    """lstm_comm's message-weight gradient  gather(h)^T D1  without the gathered copy: block (agent i, slot k) of it is
    h[nbr(i, k)]^T D1[i], and over a run of consecutive agents whose k-th neighbour sits at a constant index offset both
    operands are plain slices -- one row-split GEMM per run (3 on the line graph).  Hx / D1x [N,rows,H] contiguous."""
    key = (nbr_idx.data_ptr(), nbr_idx.device, tuple(nbr_idx.shape))
    if key not in _runs:
        tab = nbr_idx.cpu().numpy()
        runs = []
        for k in range(tab.shape[1]):
            i = 0
            while i < tab.shape[0]:
                if tab[i, k] < 0:
                    i += 1
                    continue
                d, i0 = int(tab[i, k]) - i, i
                while i < tab.shape[0] and tab[i, k] >= 0 and int(tab[i, k]) - i == d:
                    i += 1
                runs.append((i0, i, k, d))
        _runs[key] = (nbr_idx, runs)
    N, m = nbr_idx.shape
    out = torch.zeros(N, m * H, H, dtype=F32, device=Hx.device)      # slots an agent does not have: zero gradient
    for i0, i1, k, d in _runs[key][1]:
        ops.wgrad(Hx[i0 + d:i1 + d], D1x[i0:i1], out=out[i0:i1, k * H:(k + 1) * H])
    return out


def coupled_sequence_saved(kind, nbr_idx, masked_steps, enc, done, wx, wh, b, w_msg, b_msg, mfc_w, mfc_b, G, Hall, Call, S,
                           extra, s_ext=None):
    """enc [N,T,E,W] (autograd-connected h-independent part), saved activations of the rollout -> Hs [N,T,E,H].
    s_ext (optional): the [N,T+1,E,KX] buffer S is the first T slabs of (last slab zero): in-place weight gradients."""
    return CoupledSequenceSaved.apply(kind, nbr_idx, masked_steps, enc, done, wx, wh, b, w_msg, b_msg, mfc_w, mfc_b, G, Hall,
                                      Call, S, extra.get('A1'), extra.get('A2'), s_ext)


def coupled_sequence(kind, nbr_idx, masked_steps, enc, h0, c0, done, wx, wh, b, w_msg, b_msg, mfc_w=None, mfc_b=None):
    """enc [N,T,E,W] -> Hs [N,T,E,H]; see the module docstring."""
    return CoupledSequence.apply(kind, nbr_idx, masked_steps, enc, h0, c0, done, wx, wh, b, w_msg, b_msg, mfc_w, mfc_b)
