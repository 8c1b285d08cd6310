"""Host wrappers of the HIP ops of libnmarl_hip.so (include/nmarl.h): tensor
shape checks, output allocation, torch.autograd integration.  Every function
launches on torch's current HIP stream (so the rollout can be captured in a
hipGraph) and raises if given CPU tensors -- there is no fallback path.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

F32 = torch.float32


# ---- in-launch hand-off (include/nmarl.h "In-launch hand-off"): status words, capacity, the process-wide switch
_handoff_status = {}
_handoff_off = [False]


def handoff_status(device):
    """The hand-off status words of a device (int32 x 4, created once and kept -- captured hipGraphs hold the pointer):
    [0] != 0: a wave of a hand-off kernel gave up waiting (sticky until `handoff_clear`), [1] optimiser steps refused since.
    Every hand-off kernel launched through this module reports here; `rmsprop_tf_clip` consults it."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    if dev not in _handoff_status:
        _handoff_status[dev] = torch.zeros(4, dtype=torch.int32, device=dev)
    return _handoff_status[dev]


def handoff_poisoned(device):
    """True if a hand-off kernel on `device` timed out since the last `handoff_clear` (synchronises)."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        return False
    return int(handoff_status(dev)[0].item()) != 0


def handoff_skipped_updates(device):
    return int(handoff_status(device)[1].item())


def handoff_clear(device):
    handoff_status(device)[0].zero_()


def disable_inkernel_handoff():
    """Select the launch-per-step forms process-wide until `enable_inkernel_handoff` (BatchedTrainer after a time-out; same
    effect as NMARL_INKERNEL_HANDOFF=0 while it lasts)."""
    _handoff_off[0] = True


def enable_inkernel_handoff():
    """Undo `disable_inkernel_handoff` (BatchedTrainer's re-arm after a run of clean batches, a new job in the same
    process); the NMARL_INKERNEL_HANDOFF=0 environment switch still wins."""
    _handoff_off[0] = False


def handoff_enabled():
    return not _handoff_off[0] and os.environ.get('NMARL_INKERNEL_HANDOFF', '1') != '0'


def neighbor_table(neighbor_mask, device):
    """neighbor_mask [N,N] (0/1) -> (nbr_idx [N,m_max] int32 device tensor, counts list).
    Row i lists the neighbours of i in ascending index (tf.boolean_mask order), -1 padded."""
    nm = np.asarray(neighbor_mask)
    N = nm.shape[0]
    lists = [np.where(nm[i] == 1)[0] for i in range(N)]
    m_max = max(1, max(len(x) for x in lists))
    tab = -np.ones((N, m_max), dtype=np.int32)
    for i, x in enumerate(lists):
        tab[i, :len(x)] = x
    return torch.from_numpy(tab).to(device), [len(x) for x in lists]


class _NbrGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nbr_idx):
        N, E, F = x.shape
        m = nbr_idx.shape[1]
        x = x.contiguous()
        y = torch.empty(N, E, m * F, dtype=F32, device=x.device)
        check(lib.nmarl_nbr_gather_fwd(E, N, F, m, ptr(nbr_idx, torch.int32), ptr(x, F32), ptr(y), stream()),
              'nmarl_nbr_gather_fwd')
        ctx.nbr_idx = nbr_idx
        ctx.F = F
        return y

    @staticmethod
    def backward(ctx, dy):
        N, E, W = dy.shape
        m = ctx.nbr_idx.shape[1]
        dy = dy.contiguous()
        dx = torch.empty(N, E, ctx.F, dtype=F32, device=dy.device)
        check(lib.nmarl_nbr_gather_bwd(E, N, ctx.F, m, ptr(ctx.nbr_idx), ptr(dy, F32), ptr(dx), stream()),
              'nmarl_nbr_gather_bwd')
        return dx, None


class _NbrMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nbr_idx):
        N, E, F = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.nmarl_nbr_mean_fwd(E, N, F, nbr_idx.shape[1], ptr(nbr_idx, torch.int32), ptr(x, F32), ptr(y),
                                     stream()), 'nmarl_nbr_mean_fwd')
        ctx.nbr_idx = nbr_idx
        return y

    @staticmethod
    def backward(ctx, dy):
        N, E, F = dy.shape
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.nmarl_nbr_mean_bwd(E, N, F, ctx.nbr_idx.shape[1], ptr(ctx.nbr_idx), ptr(dy, F32), ptr(dx),
                                     stream()), 'nmarl_nbr_mean_bwd')
        return dx, None


def nbr_gather(x, nbr_idx):
    """x [N,E,F] -> [N,E,m_max*F]: slot k of agent i = x[nbr_idx[i,k]] (0 where padded)."""
    return _NbrGather.apply(x, nbr_idx)


def nbr_gather_bwd(dy, nbr_idx, F, add=None):
    """Adjoint of nbr_gather as a plain function (manual BPTT): dy [N,E,m_max*F] -> dx [N,E,F] (+ add [N,E,F], same pass)."""
    N, E, _ = dy.shape
    dy = dy.contiguous()
    dx = torch.empty(N, E, F, dtype=F32, device=dy.device)
    if add is not None:
        check(lib.nmarl_nbr_gather_bwd_add(E, N, F, nbr_idx.shape[1], ptr(nbr_idx, torch.int32), ptr(dy, F32), ptr(add, F32), ptr(dx),
                                           stream()), 'nmarl_nbr_gather_bwd_add')
        return dx
    check(lib.nmarl_nbr_gather_bwd(E, N, F, nbr_idx.shape[1], ptr(nbr_idx, torch.int32), ptr(dy, F32), ptr(dx), stream()),
          'nmarl_nbr_gather_bwd')
    return dx


def nbr_mean_bwd(dy, nbr_idx, add=None):
    """Adjoint of nbr_mean as a plain function: dy [N,E,F] -> dx [N,E,F] (+ add [N,E,F], same pass)."""
    N, E, F = dy.shape
    dy = dy.contiguous()
    dx = torch.empty_like(dy)
    if add is not None:
        check(lib.nmarl_nbr_mean_bwd_add(E, N, F, nbr_idx.shape[1], ptr(nbr_idx, torch.int32), ptr(dy, F32), ptr(add, F32), ptr(dx),
                                         stream()), 'nmarl_nbr_mean_bwd_add')
        return dx
    check(lib.nmarl_nbr_mean_bwd(E, N, F, nbr_idx.shape[1], ptr(nbr_idx, torch.int32), ptr(dy, F32), ptr(dx), stream()),
          'nmarl_nbr_mean_bwd')
    return dx


def nbr_mean(x, nbr_idx):
    """x [N,E,F] -> [N,E,F]: mean over the neighbours of each agent."""
    return _NbrMean.apply(x, nbr_idx)


def nbr_onehot(action, nbr_idx, n_a, out=None):
    """action [E,N] uint8 -> [N,E,m_max*A] one-hot of each agent's neighbours' actions."""
    E, N = action.shape
    m = nbr_idx.shape[1]
    if out is None:
        out = torch.empty(N, E, m * n_a, dtype=F32, device=action.device)
    if out.stride(2) != 1 or out.stride(1) != m * n_a:
        raise _lib.NmarlError('nbr_onehot: out must be [N,E,W] with contiguous [E,W] panels')
    check(lib.nmarl_nbr_onehot(E, N, n_a, m, ptr(nbr_idx, torch.int32), ptr(action, torch.uint8),
                               ptr(out, F32, strided=True), out.stride(0), stream()), 'nmarl_nbr_onehot')
    return out


def _pn(t, dtype=F32):
    """[N,E,W] tensor with contiguous [E,W] panels (e.g. slot t of an [N,T,E,W] buffer) -> (ptr, agent stride)."""
    if t is None:
        return None, 0
    if t.dim() != 3 or t.stride(2) != 1 or (t.shape[1] > 1 and t.stride(1) != t.shape[2]):
        raise _lib.NmarlError('expected [N,E,W] with contiguous [E,W] panels, got shape %s strides %s'
                              % (tuple(t.shape), t.stride()))
    return ptr(t, dtype, strided=True), t.stride(0)


def _bias(bias):
    """[N,4H] bias, possibly a strided view of the flat parameter buffer -> (ptr, row stride)."""
    if bias.stride(1) != 1:
        raise _lib.NmarlError('bias rows must be contiguous')
    return ptr(bias, F32, strided=True), bias.stride(0)


def cell_fwd(z, bias, c_prev, done, gates, c_new, h_new, z2=None):
    """Raw launch of nmarl_lstm_cell_fwd on (possibly strided) [N,E,*] panels; gates / z2 may be None."""
    N, E, H4 = z.shape
    check(lib.nmarl_lstm_cell_fwd(E, N, H4 // 4, *_pn(z), *_pn(z2), *_bias(bias), *_pn(c_prev), ptr(done, F32),
                                  *_pn(gates), *_pn(c_new), *_pn(h_new), stream()), 'nmarl_lstm_cell_fwd')


FUSED_H = 64      # nmarl_lstm_step_fused is specialised for 64-unit cells (256 gate columns per MFMA strip)


XSIDE_MAX_K = 256      # widest x-side input of nmarl_lstm_step_x (multiples of 32)


def xside_supported(kx, n_h):
    """The x-side product s @ Wx fits the fused step (csrc/lstm_mfma.hip: lstm_step_x_kernel)."""
    return n_h == FUSED_H and kx % 32 == 0 and 0 <= kx <= XSIDE_MAX_K


def lstm_wimage(wx, wh, out=None):
    """Chunked LDS image of [wx; wh] (wx [N,KX,4H] or None, wh [N,H,4H]) for the x-mode of the fused step; rebuild it
    whenever the weights change.  -> [N, (KX+64)*320] f32."""
    N = wh.shape[0]
    KX = 0 if wx is None else wx.shape[1]
    n = lib.nmarl_lstm_wimage_floats(KX)
    if out is None:
        out = torch.empty(N, n, dtype=F32, device=wh.device)
    if wh.stride(2) != 1 or wh.stride(1) != wh.shape[2] or (wx is not None and (wx.stride(2) != 1 or wx.stride(1) != wx.shape[2])):
        raise _lib.NmarlError('lstm_wimage: weights need contiguous per-agent panels')
    check(lib.nmarl_lstm_wimage(N, KX, ptr(wx, F32, strided=True), 0 if wx is None else wx.stride(0),
                                ptr(wh, F32, strided=True), wh.stride(0), ptr(out, F32), out.stride(0), stream()),
          'nmarl_lstm_wimage')
    return out


MSG_GATHER_RELU, MSG_MEAN_ADD, MSG_DIAL = 1, 2, 3      # nmarl_msg_t.kind: lstm_comm / lstm_ic3 / lstm_dial
MSG_MAX_K = 128


def msg_supported(kind, m_max, n_h):
    """The message term of a coupled net fits the step kernel's pre-phase (csrc/lstm_mfma.hip, MSG)."""
    return n_h == FUSED_H and m_max <= 8 and (n_h if kind == MSG_MEAN_ADD else n_h * m_max) <= MSG_MAX_K


def lstm_msg_wimage(w_msg, out=None):
    """LDS image of w_msg [N,K,64] for the in-kernel message term; rebuild when the weights change."""
    N, K, J = w_msg.shape
    if out is None:
        out = torch.empty(N, K * J, dtype=F32, device=w_msg.device)
    if J != FC_J or w_msg.stride(2) != 1 or w_msg.stride(1) != J:
        raise _lib.NmarlError('lstm_msg_wimage: w_msg must be [N,K,64] with contiguous panels')
    check(lib.nmarl_lstm_msg_wimage(N, K, ptr(w_msg, F32, strided=True), w_msg.stride(0), ptr(out, F32), out.stride(0), stream()),
          'nmarl_lstm_msg_wimage')
    return out


def _step_x(h, bias, zadd1, zadd2, c_prev, done, gates, c_out, h_out, xs, head, what):
    """nmarl_lstm_step_x: xs = (x [N,E,KX1] or None, wx (unused here: it is inside the image), image[, x2 [N,E,KX2][, msg]]):
    the LSTM input is [x | x2] (x2 optional), or [x | message term] with msg = dict(kind, nbr_idx, w_msg, b_msg, img,
    enc=None, out=None): the last 64 columns are computed inside the kernel from the neighbours' h (heads only); kind
    MSG_DIAL: from msg['src'] [N,E,64], the senders' message vectors, with hm (before `enc` is added) -> msg['out2']."""
    N, E, H = h.shape
    x, _, img = xs[:3]
    x2 = xs[3] if len(xs) > 3 else None
    msg = xs[4] if len(xs) > 4 else None
    if isinstance(x, dict):          # the input encoders run inside the launch (step_enc_spec): x is their description
        if head is None or head.kind != 3 or x2 is not None or msg is not None or zadd1 is not None or zadd2 is not None:
            raise _lib.NmarlError('%s: the in-kernel encoders need the policy + value step of an uncoupled net' % what)
        KX = FC_J if x.get('w_fp') is None else 2 * FC_J          # (one encoder: IA2C / ConseNet; two: IA2C-FP)
        if img.shape != (N, lib.nmarl_lstm_wimage_floats(KX)):
            raise _lib.NmarlError('%s: weight image does not match KX = %d' % (what, KX))
        check(lib.nmarl_lstm_step_x_enc(E, N, H, KX, *_pn(h), ptr(img, F32), img.stride(0), *_bias(bias), *_pn(c_prev), ptr(done, F32),
                                        *_pn(gates), *_pn(c_out), *_pn(h_out), C.byref(head), C.byref(_step_enc(x, N, E)), stream()), what)
        return
    xp, x_sn, x_row, K1 = (None, 0, 0, 0) if x is None else (*_rows_view(x, x.shape[2], what + ' x'), x.shape[2])
    x2p, x2_sn, x2_row, K2 = (None, 0, 0, 0) if x2 is None else (*_rows_view(x2, x2.shape[2], what + ' x2'), x2.shape[2])
    KX = K1 + K2 + (H if msg is not None else 0)
    if img.shape != (N, lib.nmarl_lstm_wimage_floats(KX)):
        raise _lib.NmarlError('%s: weight image does not match KX = %d' % (what, KX))
    if msg is not None:
        if head is None or x2 is not None or zadd1 is not None or zadd2 is not None:
            raise _lib.NmarlError('%s: the in-kernel message term needs a head and excludes x2 / addends' % what)
        m = _lib.Msg()
        nbr_idx = msg['nbr_idx']
        m.kind, m.m_max, m.K = msg['kind'], nbr_idx.shape[1], msg['w_msg'].shape[1]
        m.nbr_idx = ptr(nbr_idx, torch.int32)
        m.img, m.img_sn = ptr(msg['img'], F32), msg['img'].stride(0)
        m.b, m.b_sn = _bias(msg['b_msg'])
        if msg.get('enc') is not None:
            m.enc, m.enc_sn, m.enc_row = _rows_view(msg['enc'], H, what + ' enc')
        if msg.get('out') is not None:
            m.out, m.out_sn, m.out_row = _rows_view(msg['out'], H, what + ' msg out')
        if msg.get('mean_out') is not None:            # lstm_ic3, policy step: keep mean_j(h_j) for the update's message-weight gradient
            m.mean_out, m.mean_out_sn, m.mean_out_row = _rows_view(msg['mean_out'], H, what + ' msg mean_out')
        for key in ('carry_in', 'carry_out'):          # one-launch lock-steps: the re-step's message term handed to the next lock-step
            cb = msg.get(key)
            if cb is not None:
                if cb.shape != (N, E, H) or cb.stride(2) != 1 or cb.stride(1) != H:
                    raise _lib.NmarlError('%s: msg["%s"] must be [N,E,64] with contiguous panels' % (what, key))
                setattr(m, key, ptr(cb, F32, strided=True))
                setattr(m, key + '_sn', cb.stride(0))
        if msg.get('mean_next') is not None:
            m.mean_next, m.mean_next_sn, m.mean_next_row = _rows_view(msg['mean_next'], H, what + ' msg mean_next')
        if msg['kind'] == MSG_DIAL:
            src = msg['src']
            if src.shape != (N, E, H) or src.stride(2) != 1 or src.stride(1) != H:
                raise _lib.NmarlError('%s: msg["src"] must be [N,E,64] with contiguous panels' % what)
            m.src, m.src_sn = ptr(src, F32, strided=True), src.stride(0)
            if msg.get('out2') is not None:
                m.out2, m.out2_sn, m.out2_row = _rows_view(msg['out2'], H, what + ' msg out2')
            nxt = msg.get('next')          # dict(img, b, out[, w]): the sender layer on the NEW h (policy step only)
            if nxt is not None:
                m.next_img, m.next_img_sn = ptr(nxt['img'], F32), nxt['img'].stride(0)
                m.next_b, m.next_b_sn = _bias(nxt['b'])
                m.next_out, m.next_out_sn = _pn(nxt['out'])
        ob = msg.get('ob')
        if ob is not None:                     # lstm_ic3's observation encoder inside the launch (one-launch step only): writes msg['enc']
            x_ob = ob['x']                     # compact observation [E,N,F]
            if head.kind != 3 or msg['kind'] != MSG_MEAN_ADD or x_ob.dim() != 3 or x_ob.shape[:2] != (E, N) or not x_ob.is_contiguous():
                raise _lib.NmarlError('%s: the in-kernel observation encoder needs the one-launch lstm_ic3 step and a compact [E,N,F] slab' % what)
            m.ob, m.ob_row, m.ob_F, m.ob_segs = ptr(x_ob, F32), N * x_ob.shape[2], x_ob.shape[2], ob['nbr'].shape[1]
            m.ob_nbr = ptr(ob['nbr'], torch.int32)
            m.ob_img, m.ob_img_sn = ptr(ob['img'], F32), ob['img'].stride(0)
            m.ob_b, m.ob_b_sn = _bias(ob['b'])
        if head.kind == 3:                     # policy step + value re-step in one launch: the in-launch hand-off's flag words
            sync = msg.get('sync')
            if sync is None or sync.dtype != torch.int32 or sync.numel() < lib.nmarl_lstm_step_sync_words(E, N):
                raise _lib.NmarlError('%s: head kind 3 with a message term needs msg["sync"] (step_sync_words)' % what)
            m.sync = ptr(sync, torch.int32)
            m.status = ptr(handoff_status(h.device), torch.int32)
        spec = msg.get('enc_spec')
        if spec is not None:
            # NeurComm's one-launch lock-step: the two input encoders (and, with spec['env'], the env step) inside this launch; x --
            # slot t of the saved LSTM inputs -- receives their output and is read back by the K loop
            if head.kind != 3 or msg['kind'] != MSG_GATHER_RELU or x is None:
                raise _lib.NmarlError('%s: the in-kernel encoders of a coupled net need lstm_comm\'s policy + value step and its x slot' % what)
            check(lib.nmarl_lstm_step_x_msg_enc(E, N, H, KX, xp, x_sn, x_row, *_pn(h), ptr(img, F32), img.stride(0), *_bias(bias),
                                                *_pn(c_prev), ptr(done, F32), *_pn(gates), *_pn(c_out), *_pn(h_out), C.byref(head),
                                                C.byref(m), C.byref(_step_enc(dict(spec, out=None), N, E)), stream()), what)
            return
        genv = msg.get('genv')
        if genv is not None:
            # lstm_ic3 on the synthetic grid: the env step as a role of this launch (LargeGridBatchEnv.inkernel_step), on the compute
            # units the LSTM blocks leave idle
            if head.kind != 3 or msg['kind'] != MSG_MEAN_ADD or ob is None:
                raise _lib.NmarlError('%s: the in-launch grid env step needs lstm_ic3\'s one-launch step with its observation encoder inside' % what)
            g = _lib.GridEnv()
            g.params = C.pointer(genv['params'])
            for k in ('q', 'transit', 'xi', 'obs_out', 'reward', 'global_reward'):
                setattr(g, k, ptr(genv[k], F32))
            g.prev_action, g.done = ptr(genv['prev_action'], torch.uint8), ptr(genv['done'], torch.uint8)
            g.t, g.episode = ptr(genv['t'], torch.int32), ptr(genv['episode'], torch.int32)
            g.auto_reset, g.seed, g.env_id_base = (1 if genv['auto_reset'] else 0), int(genv['seed']), int(genv['env_id_base'])
            if genv['words'].dtype != torch.int64 or genv['words'].numel() < lib.nmarl_lstm_step_grid_words(E):
                raise _lib.NmarlError('%s: genv["words"] must hold nmarl_lstm_step_grid_words(E) 64-bit words' % what)
            g.words = ptr(genv['words'], torch.int64)
            check(lib.nmarl_lstm_step_x_msg_grid(E, N, H, KX, xp, x_sn, x_row, *_pn(h), ptr(img, F32), img.stride(0), *_bias(bias),
                                                 *_pn(c_prev), ptr(done, F32), *_pn(gates), *_pn(c_out), *_pn(h_out), C.byref(head),
                                                 C.byref(m), C.byref(g), stream()), what)
            return
        check(lib.nmarl_lstm_step_x_msg(E, N, H, KX, xp, x_sn, x_row, *_pn(h), ptr(img, F32), img.stride(0), *_bias(bias),
                                        *_pn(c_prev), ptr(done, F32), *_pn(gates), *_pn(c_out), *_pn(h_out), C.byref(head),
                                        C.byref(m), stream()), what)
        return
    check(lib.nmarl_lstm_step_x(E, N, H, KX, xp, x_sn, x_row, K2, x2p, x2_sn, x2_row, *_pn(h), ptr(img, F32), img.stride(0),
                                *_bias(bias), *_pn(zadd1), *_pn(zadd2), *_pn(c_prev), ptr(done, F32), *_pn(gates),
                                *_pn(c_out), *_pn(h_out), None if head is None else C.byref(head), stream()), what)


def step_enc1_supported(n_feat, m_max, n_fc, n_h, N):
    """The observation encoder ALONE fits the lock-step kernel's pre-phase (csrc/lstm_mfma.hip, ENC 2): 5 own features x (1 + m_max
    neighbours), m_max = 2 (IA2C on CACC) or 0 (ConseNet: own features only) -> 64 outputs."""
    return n_feat == 5 and m_max in (0, 2) and n_fc == FC_J and n_h == FUSED_H and N <= 32 and \
        os.environ.get('NMARL_INKERNEL_ENCODE', '1') != '0'


def step_enc_supported(n_feat, n_a, m_max, n_fc, n_h, N):
    """The two input encoders of IA2C-FP / NeurComm fit the lock-step kernel's register-only pre-phase (csrc/lstm_mfma.hip, ENC): the
    CACC input layout -- 5 own features x (1 + 2 neighbours) and 2 x 4 fingerprint entries -> 64 + 64 outputs."""
    return n_feat == 5 and n_a == 4 and m_max == 2 and n_fc == FC_J and n_h == FUSED_H and N <= 32 and \
        os.environ.get('NMARL_INKERNEL_ENCODE', '1') != '0'


def step_enc_spec(ob, fp, w_ob, b_ob, w_fp, b_fp, nbrs, out=None, env=None, bits=None):
    """Description of a lock-step's input encoders for `lstm_step_policy_value(xs=(spec, wx, image))`: ob [E,N,5] the env's
    compact observation, fp [N,E,4] the previous-step policies, the four parameter tensors as they are, nbrs = the HOST
    neighbour lists (ascending), out [N,E,128] (a view: slot t of the saved LSTM inputs) or None.
    env (optional): the CACC env step of this lock-step inside the launch as well -- CACCBatchEnv.inkernel_step(...).
    bits (optional): [N,E,4] int32 -- which of a row's 128 outputs are > 0 (layout: relu_bits_pack), for fc_concat(bits=)."""
    return dict(ob=ob, fp=fp, w_ob=w_ob, b_ob=b_ob, w_fp=w_fp, b_fp=b_fp, nbrs=nbrs, out=out, env=env, bits=bits)


def relu_bits_pack(S):
    """The bit image of (S > 0) for S [..., 128] in the layout the lock-step kernel writes (nmarl_step_enc_t.relu_bits):
    [..., 4] int32, bit 4 t + i of word q <=> S[..., 16 t + 4 q + i] > 0.  (Tests and the CPU emulation; the product's
    image comes out of the kernel.)"""
    pos = (S > 0).reshape(*S.shape[:-1], 8, 4, 4).to(torch.int64)              # [.., t, q, i]
    sh = (4 * torch.arange(8, device=S.device).view(8, 1, 1) + torch.arange(4, device=S.device).view(1, 1, 4))
    w = (pos << sh).sum(dim=(-3, -1))                                            # [.., q]
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def step_env_supported():
    return os.environ.get('NMARL_INKERNEL_ENV', '1') != '0'


_env_scratch = {}


def step_env_scratch(N, E, device):
    """The hand-off words of the in-launch env step (one per replica): per (device, N, E), zeroed once, never dropped (captured
    graphs hold the pointer; the kernel leaves them zero)."""
    key = (torch.device(device), N, E)
    if key not in _env_scratch:
        _env_scratch[key] = torch.zeros(lib.nmarl_lstm_step_env_words(E), dtype=torch.int32, device=device)
    return _env_scratch[key]


def _step_enc(d, N, E):
    ob, fp = d['ob'], d['fp']
    if ob.dim() != 3 or ob.shape != (E, N, 5) or not ob.is_contiguous():
        raise _lib.NmarlError('step_enc: ob must be the compact observation [E,N,5], contiguous')
    if d.get('w_fp') is not None and (fp.shape != (N, E, 4) or fp.stride(2) != 1 or fp.stride(1) != 4):
        raise _lib.NmarlError('step_enc: fp must be [N,E,4] with contiguous panels')
    e = _lib.StepEnc()
    e.ob, e.ob_row = ptr(ob, F32), N * 5
    single = d.get('w_fp') is None                 # the observation encoder alone (ENC 2): w_ob [N,15,64] (IA2C) or [N,5,64] (ConseNet)
    m_enc = 2 if (not single or d['w_ob'].shape[1] == 15) else 0
    if not single:
        e.fp, e.fp_sn = ptr(fp, F32, strided=True), fp.stride(0)
    for key, rows in (('w_ob', 5 * (1 + m_enc)),) + (() if single else (('w_fp', 8),)):
        w = d[key]
        if w.shape != (N, rows, FC_J):
            raise _lib.NmarlError('step_enc: %s must be [N,%d,64]' % (key, rows))
        p_, sn = _head_param(w, 'step_enc')
        setattr(e, key, p_)
        setattr(e, key + '_sn', sn)
    e.b_ob, e.b_ob_sn = _bias(d['b_ob'])
    if not single:
        e.b_fp, e.b_fp_sn = _bias(d['b_fp'])
    out = d.get('out')
    if out is not None:
        e.out, e.out_sn, e.out_row = _rows_view(out, FC_J if single else 2 * FC_J, 'step_enc out')
    bits = d.get('bits')
    if bits is not None:
        if bits.shape != (N, E, 4) or bits.dtype != torch.int32 or bits.stride(2) != 1 or bits.stride(1) != 4:
            raise _lib.NmarlError('step_enc: bits must be [N,E,4] int32 with contiguous panels')
        e.relu_bits, e.relu_bits_sn = ptr(bits, torch.int32, strided=True), bits.stride(0)
    e.F, e.A, e.m_max = 5, 4, m_enc
    for i in range(64):
        e.nbr[i] = -1
    for i, lst in enumerate(d['nbrs'] if m_enc else []):
        for k, j in enumerate(lst[:2]):
            e.nbr[2 * i + k] = int(j)
    ev = d.get('env')
    if ev is not None:
        e.env = C.pointer(ev['params'])
        for k in ('h', 'v', 'u', 'v0_init', 'obs_out', 'reward', 'global_reward'):
            setattr(e, k, ptr(ev[k], F32))
        e.t, e.episode = ptr(ev['t'], torch.int32), ptr(ev['episode'], torch.int32)
        e.collided, e.done = ptr(ev['collided'], torch.uint8), ptr(ev['done'], torch.uint8)
        e.auto_reset, e.seed, e.env_id_base = (1 if ev['auto_reset'] else 0), int(ev['seed']), int(ev['env_id_base'])
        e.cnt = ptr(step_env_scratch(N, E, ob.device), torch.int32)
        e._keep = ev['params']            # (the struct the pointer refers to stays alive with this one)
    return e


def ob_encoder_supported(n_feat, n_obs, n_h):
    """lstm_ic3's observation encoder fits the one-launch step's extra pre-phase: 16-byte feature pieces, <= 64 inputs."""
    return n_h == FUSED_H and n_feat % 4 == 0 and n_obs <= FC_J


def lstm_ob_wimage(w_ob, pad, out=None):
    """LDS image of W_ob [N,n_obs,64] for the in-kernel observation encoder: zero-padded to 64 rows in `pad` [N,64,64] (kept
    by the caller, rows >= n_obs stay zero), then the message image layout."""
    with torch.no_grad():      # (w_ob is a parameter: tracked, the copy would hang `pad` -- and an AccumulateGrad node on the stream of the
        pad[:, :w_ob.shape[1]].copy_(w_ob)       # first call -- onto the autograd graph; a node created on the default stream breaks captures)
    return lstm_msg_wimage(pad, out=out)


def step_grid_env_blocks(N, E):
    """Env-role blocks lstm_ic3's one-launch step has room for on the synthetic grid (0: none -- the env step stays a launch)."""
    return int(lib.nmarl_lstm_step_grid_env_blocks(E, N)) if handoff_enabled() else 0


def step_sync_words(N, E, device):
    """Flag words of the coupled nets' one-launch policy + value step (msg['sync']): zeroed once, then owned by the kernel."""
    return torch.zeros(lib.nmarl_lstm_step_sync_words(E, N), dtype=torch.int32, device=device)


def step_handoff_supported(N, E, device, K=128):
    """The coupled nets' policy step and value re-step fit ONE launch (blocks hand the new h over inside it): every block
    must be resident, i.e. N * ceil(E / 128) <= nmarl_handoff_capacity (the occupancy API's blocks per compute unit x compute
    units; K = floats per message row).  NMARL_INKERNEL_HANDOFF=0 / `disable_inkernel_handoff` keep the two launches (e.g.
    when several processes share one device: blocks of different processes are not co-resident by construction)."""
    if not handoff_enabled() or torch.device(device).type != 'cuda':
        return False
    cap = lib.nmarl_handoff_capacity(1, int(K))
    if cap < 0:
        raise _lib.NmarlError('nmarl_handoff_capacity failed')
    return N * ((E + 127) // 128) <= cap


def lstm_step_fused(h, wh, bias, zadd1, zadd2, c_prev, done, gates, c_out, h_out, xs=None):
    """(gates, c', h') = cell(zadd1 (+ zadd2) + (h*(1-done)) @ wh + bias, c_prev, done) in ONE MFMA kernel
    (H = 64).  All operands [N,E,*] panels (strided slots allowed); h_out / c_out may alias h / c_prev.
    xs = (x, wx, image): the x-side product x @ wx is computed inside as well (K = KX + 64, image from lstm_wimage;
    zadd1 / zadd2 may then be None)."""
    N, E, H = h.shape
    if xs is not None:
        _step_x(h, bias, zadd1, zadd2, c_prev, done, gates, c_out, h_out, xs, None, 'nmarl_lstm_step_x')
        return h_out, c_out
    if wh.stride(2) != 1 or wh.stride(1) != 4 * H:
        raise _lib.NmarlError('lstm_step_fused: wh must be [N,H,4H] with contiguous [H,4H] panels')
    check(lib.nmarl_lstm_step_fused(E, N, H, *_pn(h), ptr(wh, F32, strided=True), wh.stride(0), *_bias(bias),
                                    *_pn(zadd1), *_pn(zadd2), *_pn(c_prev), ptr(done, F32), *_pn(gates),
                                    *_pn(c_out), *_pn(h_out), stream()), 'nmarl_lstm_step_fused')
    return h_out, c_out


HEAD_MAX_A = 8      # widest action set the fused head epilogue supports (csrc/lstm_mfma.hip: MAXA)


def _fused_head(h, wh, bias, zadd1, zadd2, c_prev, done, c_out, h_out, head, what, xs=None, gates=None):
    N, E, H = h.shape
    if xs is not None:
        _step_x(h, bias, zadd1, zadd2, c_prev, done, gates, c_out, h_out, xs, head, what)
        return
    if gates is not None:
        raise _lib.NmarlError('%s: gates output needs the x-side mode' % what)
    if wh.stride(2) != 1 or wh.stride(1) != 4 * H:
        raise _lib.NmarlError('%s: wh must be [N,H,4H] with contiguous [H,4H] panels' % what)
    check(lib.nmarl_lstm_step_fused_head(E, N, H, *_pn(h), ptr(wh, F32, strided=True), wh.stride(0), *_bias(bias),
                                         *_pn(zadd1), *_pn(zadd2), *_pn(c_prev), ptr(done, F32), None, 0,
                                         *_pn(c_out), *_pn(h_out), C.byref(head), stream()), what)


def _head_param(w, what):
    """[N,rows,cols] parameter view with contiguous [rows,cols] panels -> (ptr, agent stride)."""
    if w.stride(2) != 1 or (w.shape[1] > 1 and w.stride(1) != w.shape[2]):
        raise _lib.NmarlError('%s: head weights need contiguous per-agent panels' % what)
    return ptr(w, F32, strided=True), w.stride(0)


def lstm_step_policy(h, wh, bias, zadd1, zadd2, c_prev, done, c_out, h_out, pi_w, pi_b, pi_out, act_out, mode,
                     u=None, seed=0, env_id_base=0, step=0, step_dev=None, xs=None, gates=None):
    """forward('p') of one lock-step in ONE kernel: the fused step (lstm_step_fused), then in its epilogue
    pi = softmax(h' @ pi_w + pi_b) -> pi_out [N,E,A] and the action draw of sample_actions -> act_out [E,N]."""
    N, E, H = h.shape
    A = pi_w.shape[2]
    hd = _lib.Head()
    hd.kind, hd.A, hd.mode = 1, A, mode
    hd.w, hd.w_sn = _head_param(pi_w, 'lstm_step_policy')
    hd.b, hd.b_sn = _bias(pi_b)
    hd.pi_out, hd.pi_sn = _pn(pi_out)
    hd.act_out, hd.u = ptr(act_out, torch.uint8), ptr(u, F32)
    hd.seed, hd.env_id_base, hd.step, hd.step_dev = seed, env_id_base, int(step), ptr(step_dev, torch.int64)
    _fused_head(h, wh, bias, zadd1, zadd2, c_prev, done, c_out, h_out, hd, 'nmarl_lstm_step_fused_head[p]', xs, gates)
    return pi_out, act_out


def lstm_step_value(h, wh, bias, zadd1, zadd2, c_prev, done, c_out, h_out, v_w, v_b, action, nbr_idx, n_a, v_out, xs=None):
    """forward('v') of one lock-step in ONE kernel: the fused step, then v = [h', onehot(neighbour actions)] @ v_w
    + v_b -> v_out [N,E]; the one-hot rows are gathered from action [E,N] u8 (no one-hot tensor)."""
    N, E, H = h.shape
    hd = _lib.Head()
    hd.kind, hd.A, hd.m_max = 2, n_a, nbr_idx.shape[1]
    if v_w.shape[1] != H + nbr_idx.shape[1] * n_a or v_w.shape[2] != 1:
        raise _lib.NmarlError('lstm_step_value: v_w must be [N,H+m_max*A,1]')
    hd.w, hd.w_sn = _head_param(v_w, 'lstm_step_value')
    hd.b, hd.b_sn = _bias(v_b)
    hd.act_in, hd.nbr_idx = ptr(action, torch.uint8), ptr(nbr_idx, torch.int32)
    if v_out.dim() != 2 or v_out.stride(1) != 1:
        raise _lib.NmarlError('lstm_step_value: v_out must be [N,E] with unit column stride')
    hd.v_out, hd.v_sn = ptr(v_out, F32, strided=True), v_out.stride(0)
    _fused_head(h, wh, bias, zadd1, zadd2, c_prev, done, c_out, h_out, hd, 'nmarl_lstm_step_fused_head[v]', xs)
    return v_out


def lstm_step_policy_value(h, wh, bias, zadd1, zadd2, c, done, pi_w, pi_b, pi_out, act_out, v_w, v_b, nbr_idx, n_a,
                           v_out, mode, u=None, seed=0, env_id_base=0, step=0, step_dev=None, xs=None, h_out=None,
                           c_out=None, gates=None, defer_action_term=False):
    """forward('p') AND forward('v') of one lock-step (quirk Q1) for nets without a cross-agent recurrence: one MFMA
    kernel (policy step + pi + draw, then the value re-step from the new state with the same addend and the critic on
    h'') + the critic's neighbour-action term, which needs all agents' draws, added by one small launch (unless
    `defer_action_term`: the caller adds it later for many lock-steps at once).  Coupled nets (xs carries a message term
    with msg['sync'], see step_handoff_supported): the re-step's message term comes from the neighbours' NEW h, handed over
    between the blocks inside the launch; h_out must not alias h.  The state (h, c) [N,E,64] is advanced
    by the policy step only -- in place, or into (h_out, c_out) (slots of the update's sequence buffers); `gates`
    [N,E,4H] receives the policy step's gates (x-side mode only).  v_out [N,E] with unit column stride."""
    N, E, H = h.shape
    hd = _lib.Head()
    hd.kind, hd.A, hd.mode = 3, pi_w.shape[2], mode
    hd.w, hd.w_sn = _head_param(pi_w, 'lstm_step_policy_value')
    hd.b, hd.b_sn = _bias(pi_b)
    hd.pi_out, hd.pi_sn = _pn(pi_out)
    hd.act_out, hd.u = ptr(act_out, torch.uint8), ptr(u, F32)
    hd.seed, hd.env_id_base, hd.step, hd.step_dev = seed, env_id_base, int(step), ptr(step_dev, torch.int64)
    hd.w2, hd.w2_sn = _head_param(v_w, 'lstm_step_policy_value')
    hd.b2, hd.b2_sn = _bias(v_b)
    if v_out.shape != (N, E) or v_out.stride(1) != 1 or (N > 1 and v_out.stride(0) < E):
        raise _lib.NmarlError('lstm_step_policy_value: v_out must be [N,E] with unit column stride')
    hd.v_out, hd.v_sn = ptr(v_out, F32, strided=True), v_out.stride(0)
    h_out, c_out = (h if h_out is None else h_out), (c if c_out is None else c_out)
    if xs is not None:
        _step_x(h, bias, zadd1, zadd2, c, done, gates, c_out, h_out, xs, hd, 'nmarl_lstm_step_x[pv]')
    else:
        if gates is not None:
            raise _lib.NmarlError('lstm_step_policy_value: gates output needs the x-side mode')
        _fused_head(h, wh, bias, zadd1, zadd2, c, done, c_out, h_out, hd, 'nmarl_lstm_step_fused_head[pv]')
    if not defer_action_term:
        if not v_out.is_contiguous():
            raise _lib.NmarlError('lstm_step_policy_value: the neighbour-action add needs a contiguous v_out')
        nbr_action_value(act_out, nbr_idx, v_w[:, H:], n_a, out=v_out, accumulate=True)
    return pi_out, act_out, v_out


BIAS_NONE, BIAS_RELU, BIAS_TANH = 0, 1, 2


def bias_act_(x, bias, act, out=None):
    """x = act(x + bias[:, None, :]) for x [N,rows,W] (no autograd: rollout only).  In place by default;
    `out` [N,rows,W] may be a column block of a wider [N,rows,Wtot] buffer (concat without a copy)."""
    N, rows, W = x.shape
    xp, xs = _pn(x)
    bp, bs = _bias(bias)
    if out is None:
        yp, ys, yrow = xp, xs, W
    else:
        if out.stride(2) != 1 or out.shape != x.shape:
            raise _lib.NmarlError('bias_act_: out must be [N,rows,W] with unit column stride')
        yp, ys, yrow = ptr(out, F32, strided=True), out.stride(0), out.stride(1)
    check(lib.nmarl_bias_act(rows, N, W, xp, xs, bp, bs, act, yp, ys, yrow, stream()), 'nmarl_bias_act')
    return x if out is None else out


FC_MAX_F, FC_J = 64, 64          # csrc/fc.hip: input widths up to 64, exactly 64 outputs


def _rows_view(t, W, what):
    """[N,rows,W] view with unit column stride (any agent stride / row pitch) -> (ptr, agent stride, row pitch)."""
    if t.dim() != 3 or t.shape[2] != W or t.stride(2) != 1:
        raise _lib.NmarlError('%s: expected [N,rows,%d] with unit column stride, got %s / %s' % (what, W, tuple(t.shape), t.stride()))
    return ptr(t, F32, strided=True), t.stride(0), t.stride(1)


def fc_supported(x, w):
    return x.shape[2] <= FC_MAX_F and w.shape[2] == FC_J


def fc_fwd(x, w, b, act, out=None):
    """act(x @ w + b) for x [N,rows,F<=64] (strided views allowed, e.g. the env-major slab transposed), w [N,F,64],
    b [N,64] -> out [N,rows,64] (may be a column block of a wider buffer).  No autograd."""
    N, rows, F = x.shape
    if out is None:
        out = torch.empty(N, rows, FC_J, dtype=F32, device=x.device)
    xp, xs, xr = _rows_view(x, F, 'fc_fwd x')
    yp, ys, yr = _rows_view(out, FC_J, 'fc_fwd out')
    wp, ws = _head_param(w, 'fc_fwd')
    check(lib.nmarl_fc_fwd(rows, N, F, w.shape[2], xp, xs, xr, wp, ws, *_bias(b), act, yp, ys, yr, stream()), 'nmarl_fc_fwd')
    return out


def onehot_argmax_add_(y, p, scale=None):
    """y[n,r,argmax_a p[n,r,a]] += scale[n] (1 without scale), in place: lstm_dial's own-action term one_hot(argmax(p_i), n_h)
    (agents/utils.py:577) added to the encoded observation y [N,rows,W] (a view; p [N,rows,A] with contiguous panels)."""
    N, rows, W = y.shape
    pp, p_sn = _pn(p)
    yp, ys, yr = _rows_view(y, W, 'onehot_argmax_add_ y')
    check(lib.nmarl_onehot_argmax_add(rows, N, p.shape[2], W, pp, p_sn, None if scale is None else ptr(scale, F32), yp, ys, yr, stream()),
          'nmarl_onehot_argmax_add')
    return y


def fc_fwd_multi(parts, act, out=None):
    """Several small-K layers in ONE launch: parts = [(x, w, b, nbr_idx or None), ...]; layer p writes columns
    [64p, 64p+64) of out [N,rows,64*len(parts)].  With nbr_idx the layer's input is gather(x) over the neighbour table
    (x [N,rows,A] -> F = m_max*A) without materialising it.  No autograd (rollout)."""
    N, rows = parts[0][0].shape[:2]
    n = len(parts)
    if out is None:
        out = torch.empty(N, rows, FC_J * n, dtype=F32, device=parts[0][0].device)
    arr = (_lib.FcPart * n)()
    for i, (x, w, b, nbr_idx) in enumerate(parts):
        pt = arr[i]
        pt.x, pt.x_sn, pt.x_row = _rows_view(x, x.shape[2], 'fc_fwd_multi x')
        if nbr_idx is None:
            pt.F = x.shape[2]
        else:
            pt.gather_A, pt.m_max, pt.F = x.shape[2], nbr_idx.shape[1], x.shape[2] * nbr_idx.shape[1]
            pt.nbr_idx = ptr(nbr_idx, torch.int32)
        if w.shape[1] != pt.F or w.shape[2] != FC_J:
            raise _lib.NmarlError('fc_fwd_multi: layer %d weight shape %s does not match input width %d' % (i, tuple(w.shape), pt.F))
        pt.w, pt.w_sn = _head_param(w, 'fc_fwd_multi')
        pt.b, pt.b_sn = _bias(b)
    yp, ys, yr = _rows_view(out, FC_J * n, 'fc_fwd_multi out')
    check(lib.nmarl_fc_fwd_multi(rows, N, n, arr, act, yp, ys, yr, stream()), 'nmarl_fc_fwd_multi')
    return out


def fc_bwd(x, y, dy, act, nbr_idx=None):
    """(dw [N,F,64], db [N,64]) of y = act(x @ w + b) given the layer OUTPUT y and dL/dy (column-block views allowed).
    nbr_idx [N,m_max]: the layer's input is gather(x) over the neighbour table (x [*,rows,A] -> F = m_max*A), read in place."""
    N, rows = y.shape[:2]
    F = x.shape[2] if nbr_idx is None else x.shape[2] * nbr_idx.shape[1]
    xp, xs, xr = _rows_view(x, x.shape[2], 'fc_bwd x')
    yp, ys, yr = _rows_view(y, FC_J, 'fc_bwd y')
    gp, gs, gr = _rows_view(dy, FC_J, 'fc_bwd dy')
    C_ = lib.nmarl_fc_bwd_chunks(rows, N)
    partial = torch.empty(N, C_, F + 1, FC_J, dtype=F32, device=x.device)
    dw = torch.empty(N, F, FC_J, dtype=F32, device=x.device)
    db = torch.empty(N, FC_J, dtype=F32, device=x.device)
    if nbr_idx is not None:
        check(lib.nmarl_fc_bwd_gather(rows, N, x.shape[2], nbr_idx.shape[1], ptr(nbr_idx, torch.int32), FC_J, xp, xs, xr, yp, ys, yr,
                                      gp, gs, gr, act, ptr(partial), ptr(dw), F * FC_J, ptr(db), FC_J, stream()), 'nmarl_fc_bwd_gather')
        return dw, db
    check(lib.nmarl_fc_bwd(rows, N, F, FC_J, xp, xs, xr, yp, ys, yr, gp, gs, gr, act, ptr(partial), ptr(dw), F * FC_J,
                           ptr(db), FC_J, stream()), 'nmarl_fc_bwd')
    return dw, db


def fc_bwd_pair_supported(xs, idxs, S, dS):
    """Both layers of a two-part encoding in one pass (nmarl_fc_bwd_pair): two parts with <= 16 inputs each, 16-byte rows."""
    if len(xs) != 2 or os.environ.get('NMARL_FC_BWD_PAIR', '1') == '0':
        return False
    for x, idx in zip(xs, idxs):
        if (x.shape[2] if idx is None else x.shape[2] * idx.shape[1]) > 16 or x.stride(2) != 1:
            return False
    for t in (S, dS):
        if t.shape[2] != 2 * FC_J or t.stride(2) != 1 or t.stride(1) % 4 or t.stride(0) % 4 or t.data_ptr() % 16:
            return False
    return True


def fc_bwd_pair(xs, idxs, S, dS, act, bits=None):
    """[(dw_0, db_0), (dw_1, db_1)] of S = [act(x_0 w_0 + b_0) | act(x_1 w_1 + b_1)] in ONE pass over dS [N,rows,128]: the sums
    are formed exactly as two fc_bwd calls form them.  bits [N,rows,4] int32 (relu only): the sign image of S its producer
    wrote (step_enc_spec(bits=)) -- S is then not read at all."""
    N, rows = dS.shape[:2]
    arr = (_lib.FcPart * 2)()
    Fs = []
    for i, (x, idx) in enumerate(zip(xs, idxs)):
        pt = arr[i]
        pt.x, pt.x_sn, pt.x_row = _rows_view(x, x.shape[2], 'fc_bwd_pair x')
        if idx is None:
            pt.F = x.shape[2]
        else:
            pt.gather_A, pt.m_max, pt.F = x.shape[2], idx.shape[1], x.shape[2] * idx.shape[1]
            pt.nbr_idx = ptr(idx, torch.int32)
        Fs.append(pt.F)
    gp, gs, gr = _rows_view(dS, 2 * FC_J, 'fc_bwd_pair dS')
    if bits is not None:
        if act != BIAS_RELU or bits.shape != (N, rows, 4) or bits.dtype != torch.int32 or not bits.is_contiguous():
            raise _lib.NmarlError('fc_bwd_pair: bits must be a contiguous [N,rows,4] int32 image of a relu layer')
        yp, ys, yr, bp, bs = None, 0, 0, ptr(bits, torch.int32), rows * 4
    else:
        (yp, ys, yr), bp, bs = _rows_view(S, 2 * FC_J, 'fc_bwd_pair S'), None, 0
    C_ = lib.nmarl_fc_bwd_chunks(rows, N)
    partial = torch.empty(N, C_, 2, 17, FC_J, dtype=F32, device=dS.device)
    dwb = torch.empty(N, 2, 17, FC_J, dtype=F32, device=dS.device)
    check(lib.nmarl_fc_bwd_pair(rows, N, arr, yp, ys, yr, bp, bs, gp, gs, gr, act, ptr(partial), ptr(dwb), stream()), 'nmarl_fc_bwd_pair')
    return [(dwb[:, i, :F], dwb[:, i, 16]) for i, F in enumerate(Fs)]


def _fc_concat_backward(ctx, S, xs, dS, bits=None):
    if dS.stride(2) != 1:
        dS = dS.contiguous()
    if fc_bwd_pair_supported(xs, ctx.idxs, S, dS):
        return [g for dw, db in fc_bwd_pair(xs, ctx.idxs, S, dS, ctx.act, bits=bits) for g in (None, dw, db, None)]
    grads = []
    for i, x in enumerate(xs):
        dw, db = fc_bwd(x, S[:, :, i * FC_J:(i + 1) * FC_J], dS[:, :, i * FC_J:(i + 1) * FC_J], ctx.act, nbr_idx=ctx.idxs[i])
        grads += [None, dw, db, None]
    return grads


class _FcConcat(torch.autograd.Function):
    """S = [act(x_1 w_1 + b_1) | act(x_2 w_2 + b_2) | ...]  (tf.concat of per-input fc layers, policies.py:176-181,
    agents/utils.py:186-199) for DATA inputs x_i (no dx): each block is written in place into S, and the backward
    streams S and dS once per block (fc_bwd) instead of relu-mask + skinny wgrad GEMM + bias reduction.
    args = (x_i, w_i, b_i, nbr_idx_i or None) per layer: with a table the layer's input is gather(x_i), read in place."""

    @staticmethod
    def forward(ctx, act, *args):
        xs, ws, bs, idxs = args[0::4], args[1::4], args[2::4], args[3::4]
        N, rows = ws[0].shape[0], xs[0].shape[1]
        S = torch.empty(N, rows, FC_J * len(xs), dtype=F32, device=xs[0].device)
        if any(i is not None for i in idxs):
            fc_fwd_multi(list(zip(xs, ws, bs, idxs)), act, out=S)
        else:
            for i, (x, w, b) in enumerate(zip(xs, ws, bs)):
                fc_fwd(x, w, b, act, out=S[:, :, i * FC_J:(i + 1) * FC_J])
        ctx.act, ctx.idxs = act, idxs
        ctx.save_for_backward(S, *xs)
        return S

    @staticmethod
    def backward(ctx, dS):
        S, xs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        return tuple([None] + _fc_concat_backward(ctx, S, xs, dS))


class _FcConcatSaved(torch.autograd.Function):
    """_FcConcat whose output S the rollout already computed (same weights, same inputs): no forward work.  bits: the sign
    image of S (relu layers), when the rollout's kernel wrote one."""

    @staticmethod
    def forward(ctx, act, S, bits, *args):
        xs = args[0::4]
        ctx.act, ctx.idxs, ctx.bits = act, args[3::4], bits
        ctx.save_for_backward(S, *xs)
        return S.view_as(S)

    @staticmethod
    def backward(ctx, dS):
        S, xs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        return tuple([None, None, None] + _fc_concat_backward(ctx, S, xs, dS, bits=ctx.bits))


def fc_concat(parts, act, saved=None, bits=None):
    """parts: [(x_i [N,rows,F_i], w_i [N,F_i,64], b_i [N,64][, nbr_idx_i]), ...] with data inputs -> [N,rows,64*len(parts)]
    (differentiable w.r.t. w_i, b_i).  nbr_idx_i [N,m_max] (optional): layer i's input is gather(x_i) over the neighbour
    table, x_i [*,rows,A] read in place (the env's compact observation, the fingerprints).  Inputs wider than 64 or layers
    not 64 wide: plain batched GEMMs.
    saved: the output as the rollout computed it with the current weights -- only the backward is set up (bits: with it, the
    [N,rows,4] int32 sign image of that output, relu_bits_pack's layout: the backward then does not read the output)."""
    parts = [tuple(pt) + (None,) * (4 - len(pt)) for pt in parts]
    if all((idx is not None or fc_supported(x, w)) and w.shape[1] <= FC_MAX_F and w.shape[2] == FC_J and not x.requires_grad
           for x, w, _, idx in parts):
        flat = [t for part in parts for t in part]
        if saved is not None:
            return _FcConcatSaved.apply(act, saved, bits, *flat)
        return _FcConcat.apply(act, *flat)
    f = {BIAS_NONE: lambda t: t, BIAS_RELU: torch.relu, BIAS_TANH: torch.tanh}[act]
    ys = [f(torch.baddbmm(b.unsqueeze(1), x if idx is None else nbr_gather(x, idx), w)) for x, w, b, idx in parts]
    return ys[0] if len(ys) == 1 else torch.cat(ys, dim=-1)


THIN_MAX_O = 8


def thin_linear_bwd(h, dy, w, dy2=None):
    """Backward of y = h @ w + b for h [N,rows,64], w [N,64,O<=8] -> (dh, dw [N,64,O], db [N,O]).  dL/dy is dy
    [N,rows,O], or dy [N,rows,O-1] for the leading columns plus dy2 [N,rows] for the last one."""
    N, rows, H = h.shape
    O = w.shape[2]
    if h.stride(2) != 1 or h.stride(1) != H:       # agent-strided panels are read in place (the sequence buffer's slots)
        h = h.contiguous()
    dy = dy.contiguous()
    if dy.shape[2] + (0 if dy2 is None else 1) != O:
        raise _lib.NmarlError('thin_linear_bwd: dy columns do not add up to w columns')
    C_ = lib.nmarl_fc_bwd_chunks(rows, N)
    partial = torch.empty(N, C_, H + 1, O, dtype=F32, device=h.device)
    dh = torch.empty(N, rows, H, dtype=F32, device=h.device)
    dw = torch.empty(N, H, O, dtype=F32, device=h.device)
    db = torch.empty(N, O, dtype=F32, device=h.device)
    wp, ws = _head_param(w, 'thin_linear_bwd')
    d2 = None if dy2 is None else dy2.contiguous()
    check(lib.nmarl_thin_linear_bwd(rows, N, H, O, ptr(h, F32, strided=True), h.stride(0), ptr(dy, F32), rows * dy.shape[2], ptr(d2, F32), rows,
                                    wp, ws, ptr(partial), ptr(dh), rows * H, ptr(dw), H * O, ptr(db), O, stream()),
          'nmarl_thin_linear_bwd')
    return dh, dw, db


class _ThinLinear(torch.autograd.Function):
    """y = h @ w + b with few outputs: library GEMM forward, one streaming HIP pass backward."""

    @staticmethod
    def forward(ctx, h, w, b):
        ctx.save_for_backward(h, w)
        return torch.baddbmm(b.unsqueeze(1), h, w)

    @staticmethod
    def backward(ctx, dy):
        h, w = ctx.saved_tensors
        return thin_linear_bwd(h, dy, w)


def thin_linear(h, w, b):
    """h [N,rows,H] @ w [N,H,O] + b [N,O]; H = 64 and O <= 8 take the fused backward."""
    if h.shape[2] == FC_J and w.shape[2] <= THIN_MAX_O and w.stride(2) == 1 and w.stride(1) == w.shape[2]:
        return _ThinLinear.apply(h, w, b)
    return torch.baddbmm(b.unsqueeze(1), h, w)


NBR_ACT_MAX_W = 32


def nbr_action_value(action, nbr_idx, w_a, n_a, out=None, accumulate=False):
    """va [N,rows] = onehot(neighbours' actions) @ w_a without the one-hot: action [rows,N] u8, w_a [N,m_max*A,(1)].
    `out` [N,rows] contiguous receives the result, or has it ADDED when `accumulate`."""
    rows, N = action.shape
    va = torch.empty(N, rows, dtype=F32, device=action.device) if out is None else out
    wp, ws = _head_param(w_a.reshape(N, -1, 1) if w_a.dim() == 2 else w_a, 'nbr_action_value')
    check(lib.nmarl_nbr_action_value_fwd(rows, N, n_a, nbr_idx.shape[1], ptr(nbr_idx, torch.int32), ptr(action, torch.uint8),
                                         wp, ws, ptr(va, F32), 1 if accumulate else 0, stream()), 'nmarl_nbr_action_value_fwd')
    return va


def nbr_action_value_bwd(action, nbr_idx, dv, n_a):
    """dw_a [N,m_max*A]: histogram of dv [N,rows] over (neighbour slot, action played)."""
    rows, N = action.shape
    W = nbr_idx.shape[1] * n_a
    C_ = lib.nmarl_fc_bwd_chunks(rows, N)
    partial = torch.empty(N, C_, W, dtype=F32, device=action.device)
    dw = torch.empty(N, W, dtype=F32, device=action.device)
    check(lib.nmarl_nbr_action_value_bwd(rows, N, n_a, nbr_idx.shape[1], ptr(nbr_idx, torch.int32), ptr(action, torch.uint8),
                                         ptr(dv.contiguous(), F32), ptr(partial), ptr(dw), W, stream()),
          'nmarl_nbr_action_value_bwd')
    return dw


class _Heads(torch.autograd.Function):
    """Actor logits and critic value of the update for all rows (policies.py:50-77):
        logits = h @ pi_w + pi_b ;  v = [h, onehot(neighbours' actions)] @ v_w + v_b
    Forward: ONE skinny GEMM over [pi_w | v_w[:H]] (h is read once) + the gathered neighbour-action term.
    Backward: one streaming pass for dh / d[pi_w | v_w[:H]] / db (thin_linear_bwd, the two incoming gradients read
    in place) and a histogram for d v_w[H:]; the parameter gradients come out per tensor (no cat / slice nodes)."""

    @staticmethod
    def forward(ctx, h, pi_w, pi_b, v_w, v_b, action, nbr_idx, n_a):
        H = h.shape[2]
        w = torch.cat([pi_w, v_w[:, :H]], dim=2)
        out = torch.baddbmm(torch.cat([pi_b, v_b], dim=1).unsqueeze(1), h, w)
        v = out[..., n_a] + nbr_action_value(action, nbr_idx, v_w[:, H:], n_a)
        ctx.save_for_backward(h, w, action, nbr_idx)
        ctx.n_a = n_a
        return out[..., :n_a], v

    @staticmethod
    def backward(ctx, dlogits, dv):
        h, w, action, nbr_idx = ctx.saved_tensors
        A = ctx.n_a
        dv = dv.contiguous()
        dh, dw, db = thin_linear_bwd(h, dlogits, w, dy2=dv)
        dwa = nbr_action_value_bwd(action, nbr_idx, dv, A)
        dv_w = torch.cat([dw[:, :, A:], dwa.unsqueeze(-1)], dim=1)
        return dh, dw[:, :, :A], db[:, :A], dv_w, db[:, A:], None, None, None


def heads_loss_supported(h, n_a, nbr_idx):
    """The update's heads + loss + heads' backward as ONE pass over h (nmarl_heads_loss): H = 64, A + 1 <= 8 outputs."""
    return h.is_cuda and heads_supported(h, n_a, nbr_idx) and a2c_loss_supported(n_a) and \
        os.environ.get('NMARL_FUSED_HEADS_LOSS', '1') != '0'


def heads_loss(h, pi_w, pi_b, v_w, v_b, action, nbr_idx, n_a, adv, R, v_coef, e_coef, want_dh=True):
    """h [N,rows,64] (no autograd: the caller is the root of the update's backward), action [rows,N] u8, adv / R [N,rows] ->
    dict(terms [N,3] (policy, value, entropy loss), dh [N,rows,64] or None, dy8 [N,rows,8] = [d logits | d v | 0], and the head
    parameters' gradients pi_w, pi_b, v_w, v_b) for an upstream gradient of 1 on sum_n (policy + value + entropy) -- policies.py:20-30,
    50-77; the values of ops.heads -> ops.a2c_loss -> backward, in one streaming pass over h."""
    N, rows, H = h.shape
    A = n_a
    O = A + 1
    dev = h.device
    with torch.no_grad():
        w = torch.cat([pi_w, v_w[:, :H]], dim=2)
        b = torch.cat([pi_b, v_b], dim=1)
        va = nbr_action_value(action, nbr_idx, v_w[:, H:], A)
        if h.stride(2) != 1 or h.stride(1) != H:
            raise _lib.NmarlError('heads_loss: h needs contiguous [rows,64] panels')
        C_ = lib.nmarl_fc_bwd_chunks(rows, N)
        partial = torch.empty(N, C_, 65 * O + 3, dtype=F32, device=dev)
        terms = torch.empty(N, 3, dtype=F32, device=dev)
        dy8 = torch.empty(N, rows, 8, dtype=F32, device=dev)
        dv = torch.empty(N, rows, dtype=F32, device=dev)
        dh = torch.empty(N, rows, H, dtype=F32, device=dev) if want_dh else None
        dw = torch.empty(N, H, O, dtype=F32, device=dev)
        db = torch.empty(N, O, dtype=F32, device=dev)
        check(lib.nmarl_heads_loss(rows, N, H, A, ptr(h, F32, strided=True), h.stride(0), ptr(w, F32), w.stride(0), ptr(b, F32), b.stride(0),
                                   ptr(va, F32), ptr(action, torch.uint8), ptr(adv.contiguous(), F32), ptr(R.contiguous(), F32),
                                   float(v_coef), float(e_coef), ptr(partial), ptr(terms), ptr(dy8), ptr(dv), ptr(dh), rows * H if want_dh else 0,
                                   ptr(dw), dw.stride(0), ptr(db), db.stride(0), stream()), 'nmarl_heads_loss')
        dwa = nbr_action_value_bwd(action, nbr_idx, dv, A)
        return dict(terms=terms, dh=dh, dy8=dy8, hw=w, pi_w=dw[:, :, :A], pi_b=db[:, :A],
                    v_w=torch.cat([dw[:, :, A:], dwa.unsqueeze(-1)], dim=1), v_b=db[:, A:])


def heads_supported(h, n_a, nbr_idx):
    return h.shape[2] == FC_J and n_a + 1 <= THIN_MAX_O and nbr_idx.shape[1] * n_a <= NBR_ACT_MAX_W


def heads(h, pi_w, pi_b, v_w, v_b, action, nbr_idx, n_a):
    """(logits [N,rows,A], v [N,rows]) from h [N,rows,64] and the env-major action bytes action [rows,N]."""
    return _Heads.apply(h, pi_w, pi_b, v_w, v_b, action, nbr_idx, n_a)


WGRAD_SPLIT = 16


def wgrad(a, g, out=None):
    """a^T g for a [N,rows,M], g [N,rows,K] -> [N,M,K]: the weight gradient of a batched layer, whose contraction
    runs over ALL rows (T*E of the update).  One GEMM per agent gives the library 8 tall-K problems (it reaches
    ~75 TFLOP/s fp32); splitting the rows into WGRAD_SPLIT slabs per agent (a free view) makes it 128 ordinary
    ones plus a fixed-order sum of the partial products (measured 1745 -> 936 us for 128 x 256, 1056 -> 532 us for
    64 x 256 at rows = 245760; tools/wgrad_split.py).  Deterministic."""
    N, rows, M = a.shape
    S = WGRAD_SPLIT
    if rows % S or rows < 64 * S or not a.is_contiguous() or not g.is_contiguous():
        r = torch.bmm(a.transpose(1, 2), g)
        return r if out is None else torch.mul(r, 1.0, out=out)
    part = torch.bmm(a.view(N * S, rows // S, M).transpose(1, 2), g.view(N * S, rows // S, g.shape[2]))
    part = part.view(N, S, M, g.shape[2])
    # out (a view of the caller's buffer): the partial sums land there directly -- no temporary + copy_ (whose contiguous case
    # is a hipMemcpyAsync, i.e. a memcpy node inside the captured update)
    return part.sum(1) if out is None else torch.sum(part, dim=1, out=out)


class _Linear(torch.autograd.Function):
    """y = x @ w for x [N,rows,M] (rows = T*E): plain GEMMs, with the row-split weight gradient (wgrad)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.bmm(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.bmm(dy, w.transpose(1, 2)) if ctx.needs_input_grad[0] else None
        return dx, wgrad(x, dy)


def linear(x, w):
    return _Linear.apply(x, w)


def cell_bwd(gates, c_prev, c_new, done, dh, dc, dz, dc_prev, dh2=None):
    N, E, H4 = gates.shape
    check(lib.nmarl_lstm_cell_bwd(E, N, H4 // 4, *_pn(gates), *_pn(c_prev), *_pn(c_new), ptr(done, F32), *_pn(dh),
                                  *_pn(dh2), *_pn(dc), *_pn(dz), *_pn(dc_prev), stream()), 'nmarl_lstm_cell_bwd')


def bptt_supported(H):
    return H == FUSED_H


def lstm_bptt_wimage(wxm, wh, out=None):
    """LDS image of [wxm; wh]^T for bptt_step: wxm [N,64,4H] (rows of the x-side weight that meet the h-dependent input
    part) or None, wh [N,H,4H]; rebuild when the weights change."""
    N = wh.shape[0]
    KM = 0 if wxm is None else wxm.shape[1]
    n = lib.nmarl_lstm_bptt_wimage_floats(KM)
    if out is None:
        out = torch.empty(N, n, dtype=F32, device=wh.device)
    if wh.stride(2) != 1 or wh.stride(1) != wh.shape[2] or (wxm is not None and (wxm.stride(2) != 1 or wxm.stride(1) != wxm.shape[2])):
        raise _lib.NmarlError('lstm_bptt_wimage: weights need contiguous per-agent panels')
    check(lib.nmarl_lstm_bptt_wimage(N, KM, ptr(wxm, F32, strided=True), 0 if wxm is None else wxm.stride(0),
                                     ptr(wh, F32, strided=True), wh.stride(0), ptr(out, F32), out.stride(0), stream()),
          'nmarl_lstm_bptt_wimage')
    return out


def bptt_step_db_parts(N, E, H, device):
    """Zeroed running partial sums of dz's columns for bptt_step's `db_part` ([N, parts, 4H]); .sum(1) at the end."""
    return torch.zeros(N, lib.nmarl_lstm_bptt_step_parts(E), 4 * H, dtype=F32, device=device)


def bptt_step(gates, c_prev, c_new, done, dh, dh2, dc, ws, dz, dc_prev, dhd, apply_keep, dx=None, mask=None, db_part=None):
    """cell_bwd + the dgrad product of one reverse step in ONE MFMA kernel (nmarl_lstm_bptt_step):
    dz, dc_prev as cell_bwd; dhd [N,E,H] = (dz @ wh^T) (* (1-done) if apply_keep); dx [N,E,64] = dz @ wxm^T, zeroed where
    mask <= 0.  ws = (wxm or None, wh, image from lstm_bptt_wimage).  db_part (bptt_step_db_parts): dz's column sums are
    added to it on the way (the bias gradient without a pass over dZ)."""
    N, E, H4 = gates.shape
    wxm, _, img = ws
    KM = 0 if wxm is None else wxm.shape[1]
    mp, m_sn, m_row = (None, 0, 0) if mask is None else _rows_view(mask, mask.shape[2], 'bptt_step mask')
    check(lib.nmarl_lstm_bptt_step_db(E, N, H4 // 4, KM, *_pn(gates), *_pn(c_prev), *_pn(c_new), ptr(done, F32), *_pn(dh), *_pn(dh2),
                                      *_pn(dc), ptr(img, F32), img.stride(0), *_pn(dz), *_pn(dc_prev), *_pn(dx), mp, m_sn, m_row,
                                      *_pn(dhd), 1 if apply_keep else 0, ptr(db_part, F32), 0 if db_part is None else db_part.stride(0),
                                      stream()), 'nmarl_lstm_bptt_step_db')


BPTT_SEQ_MAX_E = 1 << 21     # nmarl_lstm_bptt_seq addresses one (agent, step) panel with 32-bit byte offsets


# ---- the heads' dL/dh handed to the one-launch BPTT kernels as dy8 (ops.heads_loss) instead of as a [N,T,E,64] tensor: autograd
# carries a zero-stride placeholder of the right shape from `Hs.backward(gradient=...)` to the recurrence's backward, which
# recognises it and takes (dy8, hw) from here -- 32 bytes per row instead of 256 written and read again
_pending_head_dy = {}


def head_dy_placeholder(dy8, hw, like):
    """dy8 [N,rows,8], hw [N,64,O] -> a zero-stride tensor shaped like `like` to pass as its gradient; see take_head_dy."""
    ph = torch.zeros(1, dtype=F32, device=like.device).expand(like.shape)
    _pending_head_dy[like.device] = (ph, dy8, hw)
    return ph


def take_head_dy(dHs):
    """(dy8, hw) if dHs is the placeholder of head_dy_placeholder (nothing else contributed to dL/dh), else None."""
    pend = _pending_head_dy.pop(dHs.device, None)
    if pend is None or dHs.data_ptr() != pend[0].data_ptr() or any(dHs.stride()):
        return None
    return pend[1], pend[2]


def head_dy_to_dh(dy8, hw, shape):
    """The tensor the placeholder stands for: dL/dh = dy hw^T (recurrences without a dy8 form)."""
    O = hw.shape[2]
    return torch.bmm(dy8[:, :, :O], hw.transpose(1, 2)).view(shape)


def bptt_seq(G, Call, done, dHs, img, dZ, want_db=True, want_state_grad=False, head_dy=None):
    """The whole reverse recurrence in one launch (nmarl_lstm_bptt_seq): G / dZ [N,T,E,4H], Call [N,T+1,E,H], done [T,E],
    dHs [N,T,E,H] (the heads' dL/dh_t), img = lstm_bptt_wimage(None, wh).  -> (db [N,4H] or None, dh0, dc0 or None):
    every step masks the carried state by done_t (the reference's lstm does, agents/utils.py:104-105).
    head_dy = (dy8 [N,T*E,8], hw [N,64,O]) instead of dHs: the kernel forms the heads' dL/dh itself (nmarl_lstm_bptt_seq_dy)."""
    N, T, E, H4 = G.shape
    H = H4 // 4
    for x, w, what in ((G, H4, 'gates'), (dZ, H4, 'dz'), (Call, H, 'c_all')) + (((dHs, H, 'dh_ext'),) if head_dy is None else ()):
        if x.stride(3) != 1 or x.stride(2) != w:
            raise ValueError('bptt_seq: %s must have contiguous rows' % what)
    nblk = lib.nmarl_lstm_bptt_seq_blocks(E)
    part = torch.empty(N, nblk, H4, dtype=F32, device=G.device) if want_db else None
    dh0 = torch.empty(N, E, H, dtype=F32, device=G.device) if want_state_grad else None
    dc0 = torch.empty(N, E, H, dtype=F32, device=G.device) if want_state_grad else None
    if head_dy is not None:
        dy8, hw = head_dy
        if dy8.shape != (N, T * E, 8) or not dy8.is_contiguous() or hw.shape[:2] != (N, H) or not hw.is_contiguous():
            raise ValueError('bptt_seq: head_dy = (dy8 [N,T*E,8], hw [N,64,O]) contiguous')
        check(lib.nmarl_lstm_bptt_seq_dy(T, E, N, H, ptr(G, F32, strided=True), G.stride(0), G.stride(1), ptr(Call, F32, strided=True),
                                         Call.stride(0), Call.stride(1), ptr(done, F32), ptr(dy8, F32), dy8.stride(0), E * 8,
                                         ptr(hw, F32), hw.stride(0), hw.shape[2], ptr(img, F32), img.stride(0),
                                         ptr(dZ, F32, strided=True), dZ.stride(0), dZ.stride(1), ptr(part),
                                         0 if part is None else part.stride(0), *_pn(dh0), *_pn(dc0), stream()), 'nmarl_lstm_bptt_seq_dy')
        return (part.sum(dim=1) if want_db else None), dh0, dc0
    check(lib.nmarl_lstm_bptt_seq(T, E, N, H, ptr(G, F32, strided=True), G.stride(0), G.stride(1), ptr(Call, F32, strided=True),
                                  Call.stride(0), Call.stride(1), ptr(done, F32), ptr(dHs, F32, strided=True), dHs.stride(0),
                                  dHs.stride(1), ptr(img, F32), img.stride(0), ptr(dZ, F32, strided=True), dZ.stride(0),
                                  dZ.stride(1), ptr(part), 0 if part is None else part.stride(0), *_pn(dh0), *_pn(dc0),
                                  stream()), 'nmarl_lstm_bptt_seq')
    return (part.sum(dim=1) if want_db else None), dh0, dc0


COUPLED_NC, COUPLED_IC3 = 1, 2           # nmarl_bptt_coupled_t.kind: lstm_comm / lstm_ic3


def dial_adjoint_supported(m_max, H, rev):
    """nmarl_dial_msg_adjoint handles this message layer: 64 units, at most 4 slots and 4 sources per agent."""
    return H == FUSED_H and m_max <= 4 and rev is not None and rev['r_max'] <= 4


def dial_adjoint_images(w_msg, mfc_w):
    """The LDS images nmarl_dial_msg_adjoint multiplies by: of the transposed 64 x 64 blocks of w_msg [N,64 m,64] and of
    mfc_w^T [N,64,64] (nmarl_lstm_msg_wimage layout); once per update."""
    N, K, H = w_msg.shape
    wt = w_msg.reshape(N, K // H, H, H).transpose(2, 3).reshape(N, K, H).contiguous()
    return lstm_msg_wimage(wt), lstm_msg_wimage(mfc_w.transpose(1, 2).contiguous())


def dial_adjoint_bias_parts(N, E, device):
    """Zeroed running partial sums (b_msg's, b_mfc's gradient) for dial_msg_adjoint's `bias_parts`; .sum(1) each at the end."""
    parts = lib.nmarl_dial_msg_adjoint_parts(E)
    return torch.zeros(N, parts, FUSED_H, dtype=F32, device=device), torch.zeros(N, parts, FUSED_H, dtype=F32, device=device)


def dial_msg_adjoint(ds, hm, msg, dhd, w_msg, mfc_w, nbr_idx, imgs, rev, d1, d2, dh, bias_parts=None):
    """lstm_dial's message adjoint of one reverse step in ONE launch (nmarl_dial_msg_adjoint): d1 = ds * (hm > 0),
    d2 = gather_adjoint(d1 @ w_msg^T) * (msg > 0), dh = dhd + d2 @ mfc_w^T; all [N,E,64] panels.  imgs = dial_adjoint_images(w_msg,
    mfc_w), rev = reverse_neighbor_table(nbr_idx, COUPLED_NC) (w_msg / mfc_w / nbr_idx themselves: the restatement's inputs).
    bias_parts = dial_adjoint_bias_parts(...): the column sums of d1 / d2 are added to them on the way."""
    N, E, H = ds.shape
    b1, b2 = (None, None) if bias_parts is None else bias_parts
    check(lib.nmarl_dial_msg_adjoint(E, N, nbr_idx.shape[1], *_pn(ds), *_pn(hm), *_pn(msg), *_pn(dhd), ptr(imgs[0], F32), imgs[0].stride(0),
                                     ptr(imgs[1], F32), imgs[1].stride(0), ptr(rev['rev_agent'], torch.int32), ptr(rev['rev_col'], torch.int32),
                                     ptr(rev['rev_w'], F32), rev['r_row'], *_pn(d1), *_pn(d2), *_pn(dh), ptr(b1, F32),
                                     0 if b1 is None else b1.stride(0), ptr(b2, F32), 0 if b2 is None else b2.stride(0), stream()),
          'nmarl_dial_msg_adjoint')
    return dh


def lstm_bptt_msg_wimage(w_msg, out=None):
    """LDS image of w_msg [N,K,64] (K = 64 or 128) for the message adjoint inside nmarl_lstm_bptt_coupled."""
    N, K, J = w_msg.shape
    if out is None:
        out = torch.empty(N, K * J, dtype=F32, device=w_msg.device)
    if J != FUSED_H or w_msg.stride(2) != 1 or w_msg.stride(1) != J:
        raise _lib.NmarlError('lstm_bptt_msg_wimage: w_msg must be [N,K,64] with contiguous panels')
    check(lib.nmarl_lstm_bptt_msg_wimage(N, K, ptr(w_msg, F32, strided=True), w_msg.stride(0), ptr(out, F32), out.stride(0), stream()),
          'nmarl_lstm_bptt_msg_wimage')
    return out


def reverse_neighbor_table(nbr_idx, kind):
    """For every agent i the sources of the message adjoint: (agent a, first column of i's slot in a's message row, weight)
    for every (a, k) with nbr_idx[a, k] == i -- lstm_comm: slot k (64 k), weight 1; lstm_ic3: column 0, weight 1 / |nbr(a)|.
    -> dict(rev_agent, rev_col [N,r_row] int32, rev_w [N,r_row] f32 device tensors, r_max, r_row, symmetric) or None when an
    agent has more than 4 sources."""
    tab = nbr_idx.cpu().numpy()
    N, m = tab.shape
    lists = [[] for _ in range(N)]
    for a_ in range(N):
        cnt = int((tab[a_] >= 0).sum())
        for k in range(m):
            i = int(tab[a_, k])
            if i >= 0:
                lists[i].append((a_, 64 * k, 1.0) if kind == COUPLED_NC else (a_, 0, 1.0 / cnt))
    r_max = max(1, max(len(x) for x in lists))
    if r_max > 4:
        return None
    r_row = 2 if r_max <= 2 else 4
    ra, rc, rw = (np.zeros((N, r_row), dtype=t) for t in (np.int32, np.int32, np.float32))
    for i, lst in enumerate(lists):
        ra[i, :] = i
        for s_, (a_, col, w_) in enumerate(lst):
            ra[i, s_], rc[i, s_], rw[i, s_] = a_, col, w_
    nb = [set(int(j) for j in tab[i] if j >= 0) for i in range(N)]
    sym = all((i in nb[j]) for i in range(N) for j in nb[i])
    dev = nbr_idx.device
    return dict(rev_agent=torch.from_numpy(ra).to(dev), rev_col=torch.from_numpy(rc).to(dev), rev_w=torch.from_numpy(rw).to(dev),
                r_max=r_max, r_row=r_row, symmetric=sym)


def bptt_coupled_supported(kind, m_max, H, rev=None):
    """nmarl_lstm_bptt_coupled handles this recurrence: 64-unit cells, message rows of 64 or 128 floats; with the reverse table
    at hand also its fan-in (lstm_comm: at most 2 sources per agent -- the launcher has no 4-source instantiation of that
    kind; asymmetric tables can have m_max <= 2 and more sources)."""
    ok = H == FUSED_H and ((kind == COUPLED_NC and m_max <= 2) or kind == COUPLED_IC3)
    if ok and rev is not None and kind == COUPLED_NC and rev['r_max'] > 2:
        return False
    return ok


_coupled_ws = {}
_keepalive = [None]


def keepalive_begin(sink):
    """While a caller captures launches in a hipGraph: every pooled workspace handed out until `keepalive_end` is also
    appended to `sink` (a list the graph's owner keeps), so that the pool's LRU eviction cannot free memory whose address a
    captured graph replays."""
    _keepalive[0] = sink


def keepalive_end():
    _keepalive[0] = None


COUPLED_RING_MAX_BYTES = 4 << 30     # one slot per step (one-launch form) up to this size, else two slots (step-wise)
COUPLED_WS_KEEP = 2                  # workspaces (ring + flags + hand-over buffers) kept per process, least recently used dropped


def _coupled_workspace(dev, N, E, K, T):
    """Message buffer ([slots][N][E][K], zeroed ONCE: the kernel only needs finite contents; slots = T so that the
    one-launch form never re-reads an address, 2 if that would not fit COUPLED_RING_MAX_BYTES), flag words and the state
    hand-off buffers of nmarl_lstm_bptt_coupled -- allocated once per shape and kept (the update calls it every batch)."""
    key = (dev, N, E, K, T)
    w = _coupled_ws.pop(key, None)
    if w is not None:
        _coupled_ws[key] = w                  # most recently used last
    if w is None:
        while len(_coupled_ws) >= COUPLED_WS_KEEP:         # at most two shapes stay pinned (a training run uses one; tests many):
            _coupled_ws.pop(next(iter(_coupled_ws)))       # the update is launched eagerly, the allocator's stream ordering covers the reuse
        tiles = -(-E // 128)
        slots = T if T * N * E * K * 4 <= COUPLED_RING_MAX_BYTES else 2
        w = dict(ring=torch.zeros(max(slots, 2), N, E, K, dtype=F32, device=dev),
                 ws=torch.zeros(lib.nmarl_lstm_bptt_coupled_ws_words(E, N), dtype=torch.int32, device=dev),
                 dhr=torch.zeros(N, E, FUSED_H, dtype=F32, device=dev), dc=torch.zeros(N, E, FUSED_H, dtype=F32, device=dev),
                 db=torch.zeros(N, tiles, 4 * FUSED_H, dtype=F32, device=dev), dbm=torch.zeros(N, tiles, FUSED_H, dtype=F32, device=dev),
                 tiles=tiles)
        _coupled_ws[key] = w
    if _keepalive[0] is not None:
        _keepalive[0].append(w)
    return w


def bptt_coupled(kind, rev, m_max, G, Call, done, dHs, ws, wm, mask, dZ, D1, mode=0, head_dy=None):
    """The whole reverse recurrence of a coupled net's update in one launch (nmarl_lstm_bptt_coupled; step-wise launches of
    the same kernel when the grid is not resident at once): G / dZ [N,T,E,4H], Call [N,T+1,E,H], done [T,E], dHs / D1
    [N,T,E,H], mask (lstm_comm) = the saved message term hm as an [N,T,E,H] view (unit column stride); ws = (wxm, wh,
    lstm_bptt_wimage(wxm, wh)), wm = (w_msg, lstm_bptt_msg_wimage(w_msg)); rev = reverse_neighbor_table(nbr_idx, kind).
    -> (db [N,4H], dbmsg [N,H]); writes dZ and D1.  `ws['err']` semantics: see check_coupled_status().
    head_dy = (dy8 [N,T*E,8], hw [N,64,O]) instead of dHs: the kernel forms the heads' dL/dh itself (as bptt_seq)."""
    N, T, E, H4 = G.shape
    H = H4 // 4
    K = H * m_max if kind == COUPLED_NC else H
    img, img_m = ws[2], wm[1]
    if mode == 0 and not handoff_enabled():
        mode = 2                              # the device is shared with other processes: step-wise launches (see step_handoff_supported)
    for x, w_, what in ((G, H4, 'gates'), (dZ, H4, 'dz'), (Call, H, 'c_all'), (D1, H, 'd1')) + (((dHs, H, 'dh_ext'),) if head_dy is None else ()):
        if x.stride(3) != 1 or x.stride(2) != w_:
            raise ValueError('bptt_coupled: %s must have contiguous rows' % what)
    w = _coupled_workspace(G.device, N, E, K, T)
    a = _lib.BpttCoupled()
    a.kind, a.N, a.T, a.H, a.m_max, a.r_max, a.r_row, a.symmetric, a.mode = kind, N, T, H, m_max, rev['r_max'], rev['r_row'], \
        int(rev['symmetric']), int(mode)
    a.E = E
    a.gates, a.gates_sn, a.gates_st = ptr(G, F32, strided=True), G.stride(0), G.stride(1)
    a.c_all, a.c_sn, a.c_st = ptr(Call, F32, strided=True), Call.stride(0), Call.stride(1)
    a.done = ptr(done, F32)
    if head_dy is None:
        a.dh_ext, a.dh_sn, a.dh_st = ptr(dHs, F32, strided=True), dHs.stride(0), dHs.stride(1)
    else:
        dy8, hw = head_dy
        if dy8.shape != (N, T * E, 8) or not dy8.is_contiguous() or hw.shape[:2] != (N, H) or not hw.is_contiguous():
            raise ValueError('bptt_coupled: head_dy = (dy8 [N,T*E,8], hw [N,64,O]) contiguous')
        a.dy8, a.dy_sn, a.dy_st = ptr(dy8, F32), dy8.stride(0), E * 8
        a.hw, a.hw_sn, a.O = ptr(hw, F32), hw.stride(0), hw.shape[2]
    a.img, a.img_sn = ptr(img, F32), img.stride(0)
    a.img_m, a.imgm_sn = ptr(img_m, F32), img_m.stride(0)
    if kind == COUPLED_NC:
        if mask.dim() != 4 or mask.stride(3) != 1 or mask.shape[3] != H:
            raise ValueError('bptt_coupled: mask must be an [N,T,E,H] view with unit column stride')
        a.mask, a.mask_sn, a.mask_st, a.mask_row = ptr(mask, F32, strided=True), mask.stride(0), mask.stride(1), mask.stride(2)
    a.dz, a.dz_sn, a.dz_st = ptr(dZ, F32, strided=True), dZ.stride(0), dZ.stride(1)
    a.d1, a.d1_sn, a.d1_st = ptr(D1, F32, strided=True), D1.stride(0), D1.stride(1)
    a.ring, a.ring_sn, a.ring_slot, a.ring_slots = ptr(w['ring'], F32), w['ring'].stride(1), w['ring'].stride(0), w['ring'].shape[0]
    a.db_part, a.db_sn = ptr(w['db'], F32), w['db'].stride(0)
    a.dbm_part, a.dbm_sn = ptr(w['dbm'], F32), w['dbm'].stride(0)
    a.dhr_io, a.dc_io, a.io_sn = ptr(w['dhr'], F32), ptr(w['dc'], F32), w['dhr'].stride(0)
    a.ws = ptr(w['ws'], torch.int32)
    a.status = ptr(handoff_status(G.device), torch.int32)
    a.rev_agent, a.rev_col, a.rev_w = ptr(rev['rev_agent'], torch.int32), ptr(rev['rev_col'], torch.int32), ptr(rev['rev_w'], F32)
    check(lib.nmarl_lstm_bptt_coupled(C.byref(a), stream()), 'nmarl_lstm_bptt_coupled')
    return w['db'].sum(dim=1), w['dbm'].sum(dim=1)


def check_coupled_status(device=None):
    """Raises if a wave of ANY hand-off kernel launched on the device since the last `handoff_clear` gave up waiting for a
    neighbour's block (its results are invalid; the guarded optimiser step refused them).  Synchronises."""
    devs = list(_handoff_status) if device is None else [torch.device(device)]
    for dev in devs:
        if dev in _handoff_status and int(_handoff_status[dev][0].item()) != 0:
            raise _lib.NmarlError('in-launch hand-off: a wave timed out waiting for a neighbour block on %s (results invalid; '
                                  '%d optimiser steps refused)' % (dev, int(_handoff_status[dev][1].item())))


class _LstmCell(torch.autograd.Function):
    """(z [N,E,4H], bias [N,4H], c_prev [N,E,H], done [E]) -> (h_new, c_new)."""

    @staticmethod
    def forward(ctx, z, bias, c_prev, done):
        z = z.contiguous()
        c_prev = c_prev.contiguous()
        gates = torch.empty_like(z)
        c_new = torch.empty_like(c_prev)
        h_new = torch.empty_like(c_prev)
        cell_fwd(z, bias, c_prev, done, gates, c_new, h_new)
        ctx.save_for_backward(gates, c_prev, c_new, done)
        return h_new, c_new

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c_new, done = ctx.saved_tensors
        dz = torch.empty_like(gates)
        dc_prev = torch.empty_like(c_prev)
        dh = None if dh is None else dh.contiguous()
        dc = None if dc is None else dc.contiguous()
        cell_bwd(gates, c_prev, c_new, done, dh, dc, dz, dc_prev)
        return dz, dz.sum(dim=1), dc_prev, None


def lstm_cell(z, bias, c_prev, done):
    return _LstmCell.apply(z, bias, c_prev, done)


def lstm_cell_infer(z, bias, c_prev, done, c_out, h_out, z2=None):
    """No-autograd cell for the rollout: gates are not materialised; c/h go to the given buffers
    (which may alias c_prev / the previous h); z2 is an optional second pre-activation addend."""
    cell_fwd(z, bias, c_prev, done, None, c_out, h_out, z2=z2)
    return h_out, c_out


class _LstmSequence(torch.autograd.Function):
    """cuDNN-style fused recurrence for agent-batched LSTMs without cross-agent coupling
    (LstmPolicy / FPPolicy, agents/utils.py:87-115 unrolled over n_step):

        z_t = pre[:, t] + (h_{t-1} * (1 - done_t)) @ wh ;  (h_t, c_t) = cell(z_t + b, c_{t-1}, done_t)

    pre [N,T,E,4H] (x-side pre-activations, no bias), wh [N,H,4H], b [N,4H], h0/c0 [N,E,H],
    done [T,E] f32 -> Hs [N,T,E,H].  Forward keeps gates / c / h for all steps in three sequence
    buffers; backward is ONE reverse loop of (cell_bwd, dgrad GEMM) plus, after the loop, a single
    wgrad GEMM over all T*E rows and a single bias reduction -- instead of T small ones."""

    @staticmethod
    def forward(ctx, pre, wh, b, h0, c0, done, masked_steps):
        N, T, E, H4 = pre.shape
        H = H4 // 4
        G = torch.empty(N, T, E, H4, dtype=F32, device=pre.device)
        Hall = torch.empty(N, T + 1, E, H, dtype=F32, device=pre.device)
        Call = torch.empty(N, T + 1, E, H, dtype=F32, device=pre.device)
        Hall[:, 0].copy_(h0)
        Call[:, 0].copy_(c0)
        keep = (1.0 - done)                                                   # [T,E]
        masked = set(range(T)) if masked_steps is None else set(masked_steps)
        fused = H == FUSED_H and wh.stride(2) == 1 and wh.stride(1) == H4
        for t in range(T):
            if fused:     # recurrent GEMM + cell in one MFMA kernel, pre-activation never leaves the CU
                lstm_step_fused(Hall[:, t], wh, b, pre[:, t], None, Call[:, t], done[t], G[:, t], Call[:, t + 1],
                                Hall[:, t + 1])
                continue
            hk = Hall[:, t] * keep[t].view(1, E, 1) if t in masked else Hall[:, t]
            # recurrent product as a plain GEMM; the x-side pre-activation enters the cell kernel as 2nd addend
            cell_fwd(torch.bmm(hk, wh), b, Call[:, t], done[t], G[:, t], Call[:, t + 1], Hall[:, t + 1], z2=pre[:, t])
        ctx.save_for_backward(G, Hall, Call, wh, done)
        ctx.masked = masked
        return Hall[:, 1:]

    @staticmethod
    def backward(ctx, dHs):
        G, Hall, Call, wh, done = ctx.saved_tensors
        N, T, E, H4 = G.shape
        H = H4 // 4
        dHs = dHs.contiguous()
        dZ = torch.empty_like(G)
        keep = (1.0 - done)
        dh_rec = None
        dc = torch.zeros(N, E, H, dtype=F32, device=G.device)
        dc_next = torch.empty_like(dc)
        wh_t = wh.transpose(1, 2)
        for t in range(T - 1, -1, -1):
            # dL/dh_t = head part dHs[:, t] (strided slot) + recurrent part dh_rec, summed inside the kernel
            cell_bwd(G[:, t], Call[:, t], Call[:, t + 1], done[t], dHs[:, t], dc, dZ[:, t], dc_next, dh2=dh_rec)
            dc, dc_next = dc_next, dc
            dh_rec = torch.bmm(dZ[:, t], wh_t)
            if t in ctx.masked:
                dh_rec = dh_rec * keep[t].view(1, E, 1)
        dZf = dZ.view(N, T * E, H4)
        # h_{t-1} * keep_t for all t: Hall[:, :T] is a strided view (agent stride (T+1)*E*H)
        if len(ctx.masked) == T:
            Hprev = (Hall[:, :T] * keep.view(1, T, E, 1)).reshape(N, T * E, H)
        else:
            Hprev = Hall[:, :T].clone()
            for t in ctx.masked:
                Hprev[:, t].mul_(keep[t].view(1, E, 1))
            Hprev = Hprev.view(N, T * E, H)
        dwh = wgrad(Hprev, dZf)
        db = dZf.sum(dim=1)
        return dZ, dwh, db, dh_rec, dc, None, None


def lstm_sequence(pre, wh, b, h0, c0, done, masked_steps=None):
    """masked_steps: the steps t whose done[t] can be non-zero (None = any); the others skip the state-mask
    multiplies (in the batched trainer only t = 0 can start an episode, quirk Q4)."""
    return _LstmSequence.apply(pre, wh, b, h0, c0, done, masked_steps)


class _LstmSequenceX(torch.autograd.Function):
    """_LstmSequence with the x-side product INSIDE the step kernel (nmarl_lstm_step_x):

        z_t = s[:, t] @ wx + (h_{t-1} * (1 - done_t)) @ wh ;  (h_t, c_t) = cell(z_t + b, c_{t-1}, done_t)

    s [N,T,E,KX] (the encoders' output, e.g. [fcs | fcp]), wx [N,KX,4H], wh [N,H,4H]: no [N,T,E,4H] pre-activation
    tensor and no forward GEMM over T*E rows.  Backward: the reverse loop of (cell_bwd, dgrad GEMM vs wh), then
    ds = dZ @ wx^T, dwx = s^T dZ, dwh, db as single GEMMs / reductions over all T*E rows."""

    @staticmethod
    def forward(ctx, s, wx, wh, b, h0, c0, done, masked_steps, img):
        N, T, E, KX = s.shape
        H = wh.shape[1]
        H4 = 4 * H
        G = torch.empty(N, T, E, H4, dtype=F32, device=s.device)
        Hall = torch.empty(N, T + 1, E, H, dtype=F32, device=s.device)
        Call = torch.empty(N, T + 1, E, H, dtype=F32, device=s.device)
        Hall[:, 0].copy_(h0)
        Call[:, 0].copy_(c0)
        for t in range(T):
            lstm_step_fused(Hall[:, t], wh, b, None, None, Call[:, t], done[t], G[:, t], Call[:, t + 1], Hall[:, t + 1],
                            xs=(s[:, t], wx, img))
        ctx.save_for_backward(G, Hall, Call, s, wx, wh, done)
        ctx.masked = set(range(T)) if masked_steps is None else set(masked_steps)
        return Hall[:, 1:]

    @staticmethod
    def backward(ctx, dHs):
        G, Hall, Call, s, wx, wh, done = ctx.saved_tensors
        ds, dwx, dwh, db, dh_rec, dc = _lstm_seq_x_backward(G, Hall, Call, s, wx, wh, done, ctx.masked, dHs,
                                                            ctx.needs_input_grad[0])
        return ds, dwx, dwh, db, dh_rec, dc, None, None, None


def _lstm_seq_x_backward(G, Hall, Call, s, wx, wh, done, masked, dHs, need_ds, want_state_grad=True, s_ext=None):
    """BPTT of z_t = s_t @ wx + (h_{t-1} keep_t) @ wh, (h_t, c_t) = cell(z_t + b, c_{t-1}, done_t) from the saved gates
    G [N,T,E,4H] and state sequences Hall / Call [N,T+1,E,H]: the reverse loop of (cell_bwd, dgrad GEMM vs wh), then
    ds = dZ @ wx^T, dwx = s^T dZ, dwh = (h keep)^T dZ, db = sum dZ over all T*E rows."""
    N, T, E, H4 = G.shape
    H = H4 // 4
    head_dy = take_head_dy(dHs)
    one_launch = bptt_supported(H) and wh.stride(2) == 1 and wh.stride(1) == H4 and E <= BPTT_SEQ_MAX_E
    if head_dy is not None and not one_launch:
        dHs, head_dy = head_dy_to_dh(*head_dy, dHs.shape), None
    if head_dy is None:
        dHs = dHs.contiguous()
    keep = (1.0 - done)
    KX = s.shape[3]
    # In-place weight gradients: with the LSTM inputs handed over as the first T slabs of a (T + 1)-slab buffer (last slab
    # zero), dZ allocated the same way (last slab zero) and the h sequence being (T + 1) slabs anyway, all three are plain
    # contiguous [N, (T+1)*E, .] operands of the row-split GEMMs: no masked COPY of the h sequence (0.2 ms at T*E*N = 2 M
    # rows).  The few maskable steps are masked in the saved buffer itself (the rollout rewrites it next batch).
    ext = (s_ext is not None and len(masked) < T and s_ext.shape == (N, T + 1, E, KX) and s_ext.is_contiguous() and Hall.is_contiguous()
           and s.data_ptr() == s_ext.data_ptr() and ((T + 1) * E) % WGRAD_SPLIT == 0)
    if ext:
        dZe = torch.empty(N, T + 1, E, H4, dtype=F32, device=G.device)
        dZe[:, T].zero_()
        dZ = dZe[:, :T]
    else:
        dZ = torch.empty_like(G)
    db = None
    if one_launch:
        # the whole reverse recurrence in one launch; the bias gradient comes out of the same pass.  It multiplies by
        # (1 - done_t) at every step: exact also for the steps outside `masked`, whose done_t is zero by contract
        db, dh_rec, dc = bptt_seq(G, Call, done, dHs, lstm_bptt_wimage(None, wh), dZ, want_state_grad=want_state_grad, head_dy=head_dy)
    else:
        dh_rec = None
        dc = torch.zeros(N, E, H, dtype=F32, device=G.device)
        dc_next = torch.empty_like(dc)
        wh_t = wh.transpose(1, 2)
        for t in range(T - 1, -1, -1):
            cell_bwd(G[:, t], Call[:, t], Call[:, t + 1], done[t], dHs[:, t], dc, dZ[:, t], dc_next, dh2=dh_rec)
            dc, dc_next = dc_next, dc
            dh_rec = torch.bmm(dZ[:, t], wh_t)
            if t in masked:
                dh_rec = dh_rec * keep[t].view(1, E, 1)
    dZf = dZ.reshape(N, T * E, H4)              # a view in both layouts (agent-strided when `ext`)
    if ext:
        with torch.no_grad():
            for t in masked:
                Hall[:, t].mul_(keep[t].view(1, E, 1))
        dZx = dZe.view(N, (T + 1) * E, H4)
        dwh = wgrad(Hall.view(N, (T + 1) * E, H), dZx)
        dwx = wgrad(s_ext.view(N, (T + 1) * E, KX), dZx)
    else:
        if len(masked) == T:
            Hprev = (Hall[:, :T] * keep.view(1, T, E, 1)).reshape(N, T * E, H)
        else:
            Hprev = Hall[:, :T].clone()
            for t in masked:
                Hprev[:, t].mul_(keep[t].view(1, E, 1))
            Hprev = Hprev.view(N, T * E, H)
        dwh = wgrad(Hprev, dZf)
        dwx = wgrad(s.reshape(N, T * E, KX), dZf)
    if db is None:
        db = dZf.sum(dim=1)
    ds = torch.bmm(dZf, wx.transpose(1, 2)).view(N, T, E, KX) if need_ds else None
    return ds, dwx, dwh, db, dh_rec, dc


class _LstmSequenceSaved(torch.autograd.Function):
    """The recurrence of the update WITHOUT a forward pass: the rollout already evaluated exactly this sequence with
    exactly these weights (on-policy A2C: one update per batch, states_bw = the states the rollout started from), and
    its step kernel saved the gates and the state sequences.  forward = hand out Hall[:, 1:]; backward = the BPTT of
    _LstmSequenceX.  The reference recomputes the forward inside its training graph (policies.py:99-100, 330-331):
    same weights, same inputs, same function -- the values are those of the rollout."""

    @staticmethod
    def forward(ctx, s, wx, wh, b, G, Hall, Call, done, masked_steps, s_ext):
        T = G.shape[1]
        ctx.save_for_backward(G, Hall, Call, s, wx, wh, done)
        ctx.masked = set(range(T)) if masked_steps is None else set(masked_steps)
        ctx.s_ext = s_ext
        return Hall[:, 1:]

    @staticmethod
    def backward(ctx, dHs):
        G, Hall, Call, s, wx, wh, done = ctx.saved_tensors
        ds, dwx, dwh, db, _, _ = _lstm_seq_x_backward(G, Hall, Call, s, wx, wh, done, ctx.masked, dHs,
                                                      ctx.needs_input_grad[0], want_state_grad=False, s_ext=ctx.s_ext)
        return ds, dwx, dwh, db, None, None, None, None, None, None


def lstm_sequence_saved(s, wx, wh, b, G, Hall, Call, done, masked_steps, s_ext=None):
    """s [N,T,E,KX] (autograd-connected encoders' output), G / Hall / Call saved by the rollout -> Hs [N,T,E,H].
    s_ext (optional): the [N,T+1,E,KX] buffer whose first T slabs s is a view of, last slab zero: the weight gradients then
    read the saved sequences in place (see _lstm_seq_x_backward)."""
    return _LstmSequenceSaved.apply(s, wx, wh, b, G, Hall, Call, done, masked_steps, s_ext)


def lstm_sequence_x(s, wx, wh, b, h0, c0, done, masked_steps, img):
    """s [N,T,E,KX] contiguous -> Hs [N,T,E,H]; img = lstm_wimage(wx, wh) of the CURRENT weights."""
    return _LstmSequenceX.apply(s, wx, wh, b, h0, c0, done, masked_steps, img)


SAMPLE_UNIFORM, SAMPLE_PHILOX, SAMPLE_ARGMAX = 0, 1, 2


def sample_actions(pi, out, mode, u=None, seed=0, env_id_base=0, step=0, step_dev=None):
    """pi [N,E,A] -> out [E,N] uint8 (utils.py:135-141).  step_dev: optional device int64 scalar
    holding the global lock-step (Philox counter) -- used inside captured hipGraphs."""
    N, E, A = pi.shape
    check(lib.nmarl_sample_actions(E, N, A, ptr(pi, F32), ptr(u, F32), mode, seed, env_id_base, int(step),
                                   ptr(step_dev, torch.int64), ptr(out, torch.uint8), stream()),
          'nmarl_sample_actions')
    return out


A2C_LOSS_MAX_A = 8


class _A2CLoss(torch.autograd.Function):
    """(total [N], terms [N,3]) = per-agent (policy, value, entropy) loss terms of Policy.prepare_loss (policies.py:20-30) from logits
    [N,rows,A] (any row pitch), v / adv / R [N,rows] and the env-major action bytes [rows,N]: one streaming pass
    forward, one backward (closed-form d/dlogits, d/dv), instead of the softmax / log / clamp / gather / mean chain."""

    @staticmethod
    def forward(ctx, logits, v, action, adv, R, v_coef, e_coef):
        N, rows, A = logits.shape
        if logits.stride(2) != 1:
            logits = logits.contiguous()
        v, adv, R = v.contiguous(), adv.contiguous(), R.contiguous()
        C_ = lib.nmarl_a2c_loss_chunks(rows, N)
        partial = torch.empty(N, C_, 3, dtype=F32, device=logits.device)
        out = torch.empty(N, 3, dtype=F32, device=logits.device)
        check(lib.nmarl_a2c_loss_fwd(rows, N, A, ptr(logits, F32, strided=True), logits.stride(0), logits.stride(1), ptr(v, F32),
                                     ptr(action, torch.uint8), ptr(adv, F32), ptr(R, F32), v_coef, e_coef, ptr(partial), ptr(out),
                                     stream()), 'nmarl_a2c_loss_fwd')
        ctx.save_for_backward(logits, v, action, adv, R)
        ctx.coefs = (v_coef, e_coef)
        ctx.mark_non_differentiable(out)
        return out.sum(dim=1), out            # (differentiable per-agent total [N], the three terms for logging)

    @staticmethod
    def backward(ctx, g, _):
        logits, v, action, adv, R = ctx.saved_tensors
        N, rows, A = logits.shape
        gn = g.contiguous()
        dlogits = torch.empty(N, rows, A, dtype=F32, device=logits.device)
        dv = torch.empty(N, rows, dtype=F32, device=logits.device)
        check(lib.nmarl_a2c_loss_bwd(rows, N, A, ptr(logits, F32, strided=True), logits.stride(0), logits.stride(1), ptr(v, F32),
                                     ptr(action, torch.uint8), ptr(adv, F32), ptr(R, F32), ctx.coefs[0], ctx.coefs[1],
                                     ptr(gn, F32), ptr(dlogits), ptr(dv), stream()), 'nmarl_a2c_loss_bwd')
        return dlogits, dv, None, None, None, None, None


def a2c_loss_supported(n_a):
    return n_a <= A2C_LOSS_MAX_A


def a2c_loss(logits, v, action, adv, R, v_coef, e_coef):
    """-> per-agent total loss [N] (differentiable) and the detached (policy, value, entropy) terms [N,3]."""
    return _A2CLoss.apply(logits, v, action, adv, R, float(v_coef), float(e_coef))


def nstep_return(r, v, done_post, R_end, gamma, alpha, dist=None, R_out=None, adv_out=None):
    """r [T,E] | [T,E,N], v [T,N,E], done_post [T,E] u8, R_end [N,E] -> R, Adv [N,T,E]."""
    T, N, E = v.shape
    if R_out is None:
        R_out = torch.empty(N, T, E, dtype=F32, device=v.device)
    if adv_out is None:
        adv_out = torch.empty(N, T, E, dtype=F32, device=v.device)
    check(lib.nmarl_nstep_return(E, N, T, ptr(r, F32), ptr(v, F32), ptr(done_post, torch.uint8), ptr(R_end, F32),
                                 float(gamma), float(alpha), ptr(dist, torch.int32), ptr(R_out), ptr(adv_out),
                                 stream()), 'nmarl_nstep_return')
    return R_out, adv_out


def rmsprop_tf_clip(w, g, ms, scratch, lr, rho, eps, max_norm, grad_scale=1.0, norm_out=None, lr_dev=None, guard=False):
    """In-place clip_by_global_norm + TF RMSProp on flat [G,P] buffers.  guard: the caller's model launches in-launch
    hand-off kernels -- the step then consults the device's hand-off status word and changes nothing while it is set (a
    batch whose hand-off timed out cannot reach the weights: fail closed).  Models without such kernels do not look at the
    word, so another model's time-out on the same device cannot silence their updates."""
    G, P = w.shape
    check(lib.nmarl_rmsprop_tf_clip_guarded(G, P, ptr(w, F32), ptr(g, F32), ptr(ms, F32), ptr(scratch, F32),
                                            ptr(lr_dev, F32), float(lr), float(rho), float(eps), float(max_norm),
                                            float(grad_scale), ptr(norm_out, F32),
                                            ptr(handoff_status(w.device), torch.int32) if (guard and w.is_cuda) else None, stream()),
          'nmarl_rmsprop_tf_clip')


_epilogue_scratch = {}


def batch_epilogue(g, done, ep_sum, ep_sq, ep_len, fin, T_env, h_fw, c_fw, h_bw, c_bw, fp_T, fp_0, fp_uniform, x_T, x_0, done_pre,
                   skip_if=None):
    """nmarl_batch_epilogue: episode statistics + the state hand-over between two n_step batches (see include/nmarl.h).
    g [T,E] f32, done [E] u8, ep_* [E] f64, fin [4] f64; h_* / c_* [N,E,H]; fp_T / fp_0 [N,E,A], fp_uniform [N,1,A] or [N,A];
    x_T / x_0 [E,N,F]; done_pre [E] f32.  skip_if: int32 device word -- while != 0 the call changes nothing."""
    T, E = g.shape
    N, _, H = h_fw.shape
    a = _lib.BatchEpilogue()
    a.E, a.N, a.H, a.A, a.F, a.T, a.T_env = E, N, H, fp_T.shape[2], x_T.shape[2], T, int(T_env)
    a.g, a.done = ptr(g, F32), ptr(done, torch.uint8)
    a.ep_sum, a.ep_sq, a.ep_len, a.fin = (ptr(t, torch.float64) for t in (ep_sum, ep_sq, ep_len, fin))
    a.h_fw, a.c_fw, a.h_bw, a.c_bw = (ptr(t, F32) for t in (h_fw, c_fw, h_bw, c_bw))
    a.fp_T, a.fp_0, a.fp_uniform = ptr(fp_T, F32), ptr(fp_0, F32), ptr(fp_uniform.reshape(N, -1), F32)
    a.x_T, a.x_0, a.done_pre = ptr(x_T, F32), ptr(x_0, F32), ptr(done_pre, F32)
    key = g.device
    if key not in _epilogue_scratch:
        _epilogue_scratch[key] = torch.zeros(4096, dtype=torch.float64, device=g.device)      # NMARL_EPILOGUE_SCRATCH
    a.scratch = ptr(_epilogue_scratch[key], torch.float64)
    a.skip_if = ptr(skip_if, torch.int32, strided=True)
    check(lib.nmarl_batch_epilogue(C.byref(a), stream()), 'nmarl_batch_epilogue')


COPY_MAX = 16      # NMARL_COPY_MAX


def copy_multi(pairs, skip_if=None):
    """nmarl_copy_multi: [(dst, src), ...] contiguous device tensors of equal byte size, copied by ONE kernel launch per 16
    pairs on the current stream -- inside a captured hipGraph a kernel node, where `dst.copy_(src)` would be a memcpy node.
    skip_if: int32 device word -- while != 0 nothing is copied."""
    pairs = list(pairs)
    for d, s_ in pairs:
        if not (d.is_contiguous() and s_.is_contiguous()) or d.dtype != s_.dtype or d.numel() != s_.numel():
            raise _lib.NmarlError('copy_multi: pairs must be contiguous tensors of one dtype and size (got %s %s <- %s %s)'
                                  % (tuple(d.shape), d.dtype, tuple(s_.shape), s_.dtype))
    for i in range(0, len(pairs), COPY_MAX):
        part = pairs[i:i + COPY_MAX]
        n = len(part)
        dst = (C.c_void_p * n)(*[ptr(d) for d, _ in part])
        src = (C.c_void_p * n)(*[ptr(s_) for _, s_ in part])
        nbytes = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in part])
        check(lib.nmarl_copy_multi(n, dst, src, nbytes, ptr(skip_if, torch.int32, strided=True), stream()), 'nmarl_copy_multi')


def timestamp(out):
    """nmarl_timestamp: out [1] int64 device tensor <- the device's constant-rate wall clock, from a one-thread kernel on the
    current stream (capturable: a kernel node).  Measurement only."""
    if out.dtype != torch.int64 or out.numel() != 1:
        raise _lib.NmarlError('timestamp: out must be one int64')
    check(lib.nmarl_timestamp(ptr(out, torch.int64, strided=True), stream()), 'nmarl_timestamp')


def timestamp_rate_khz(device=None):
    khz = lib.nmarl_timestamp_rate_khz()
    if khz <= 0:
        raise _lib.NmarlError('nmarl_timestamp_rate_khz failed (%d)' % khz)
    return khz
