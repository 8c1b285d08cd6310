"""Host wrappers of the HIP ops of libnmarl_hip.so (include/nmarl.h): tensor
shape checks, output allocation, torch.autograd integration.  Every function
launches on torch's current HIP stream (so the rollout can be captured in a
hipGraph) and raises if given CPU tensors -- there is no fallback path.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

F32 = torch.float32


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


def _bias(bias):
    """[N,4H] bias, possibly a strided view of the flat parameter buffer -> (ptr, row stride)."""
    if bias.stride(1) != 1:
        raise _lib.NmarlError('bias rows must be contiguous')
    return ptr(bias, F32, strided=True), bias.stride(0)


class _LstmCell(torch.autograd.Function):
    """(z [N,E,4H], bias [N,4H], c_prev [N,E,H], done [E]) -> (h_new, c_new)."""

    @staticmethod
    def forward(ctx, z, bias, c_prev, done):
        N, E, H4 = z.shape
        H = H4 // 4
        z = z.contiguous()
        c_prev = c_prev.contiguous()
        gates = torch.empty_like(z)
        c_new = torch.empty_like(c_prev)
        h_new = torch.empty_like(c_prev)
        check(lib.nmarl_lstm_cell_fwd(E, N, H, ptr(z, F32), *_bias(bias), ptr(c_prev, F32), ptr(done, F32),
                                      ptr(gates), ptr(c_new), ptr(h_new), stream()), 'nmarl_lstm_cell_fwd')
        ctx.save_for_backward(gates, c_prev, c_new, done)
        return h_new, c_new

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c_new, done = ctx.saved_tensors
        N, E, H4 = gates.shape
        H = H4 // 4
        dz = torch.empty_like(gates)
        dc_prev = torch.empty_like(c_prev)
        dh = None if dh is None else dh.contiguous()
        dc = None if dc is None else dc.contiguous()
        check(lib.nmarl_lstm_cell_bwd(E, N, H, ptr(gates), ptr(c_prev), ptr(c_new), ptr(done), ptr(dh), ptr(dc),
                                      ptr(dz), ptr(dc_prev), stream()), 'nmarl_lstm_cell_bwd')
        return dz, dz.sum(dim=1), dc_prev, None


def lstm_cell(z, bias, c_prev, done):
    return _LstmCell.apply(z, bias, c_prev, done)


def lstm_cell_infer(z, bias, c_prev, done, c_out, h_out):
    """No-autograd cell for the rollout: gates are written over z, c/h into the given buffers."""
    N, E, H4 = z.shape
    check(lib.nmarl_lstm_cell_fwd(E, N, H4 // 4, ptr(z, F32), *_bias(bias), ptr(c_prev, F32), ptr(done, F32),
                                  ptr(z), ptr(c_out, F32), ptr(h_out, F32), stream()), 'nmarl_lstm_cell_fwd')
    return h_out, c_out


SAMPLE_UNIFORM, SAMPLE_PHILOX, SAMPLE_ARGMAX = 0, 1, 2


def sample_actions(pi, out, mode, u=None, seed=0, env_id_base=0, step=0, step_dev=None):
    """pi [N,E,A] -> out [E,N] uint8 (utils.py:135-141).  step_dev: optional device int64 scalar
    holding the global lock-step (Philox counter) -- used inside captured hipGraphs."""
    N, E, A = pi.shape
    check(lib.nmarl_sample_actions(E, N, A, ptr(pi, F32), ptr(u, F32), mode, seed, env_id_base, int(step),
                                   ptr(step_dev, torch.int64), ptr(out, torch.uint8), stream()),
          'nmarl_sample_actions')
    return out


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


def rmsprop_tf_clip(w, g, ms, scratch, lr, rho, eps, max_norm, grad_scale=1.0, norm_out=None, lr_dev=None):
    """In-place clip_by_global_norm + TF RMSProp on flat [G,P] buffers."""
    G, P = w.shape
    check(lib.nmarl_rmsprop_tf_clip(G, P, ptr(w, F32), ptr(g, F32), ptr(ms, F32), ptr(scratch, F32),
                                    ptr(lr_dev, F32), float(lr), float(rho), float(eps), float(max_norm),
                                    float(grad_scale), ptr(norm_out, F32), stream()), 'nmarl_rmsprop_tf_clip')
