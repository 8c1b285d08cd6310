"""Batched per-agent LSTM policies on MI355X (PyTorch-ROCm GEMMs + the HIP ops of
csrc/nbr.hip and csrc/a2c.hip), mirroring the reference's agents/policies.py.

Reference -> here (all agents evaluated at once, E replicas at once):
  LstmPolicy          policies.py:80-154   + fc/lstm      agents/utils.py:65-115
  FPPolicy            policies.py:157-185
  NCMultiAgentPolicy  policies.py:188-336  + lstm_comm    agents/utils.py:118-217
  IC3MultiAgentPolicy policies.py:429-476  + lstm_ic3     agents/utils.py:344-417
  heads               policies.py:50-77

Layout.  Every agent owns its own weights (nothing is shared), so each layer is
ONE batched GEMM with batch = agent: activations are agent-major [N, E, F]
(rollout) / [N, T*E, F] (update), weights [N, F_in, F_out].  Agents with fewer
neighbours than m_max are zero padded: their inputs for the missing slots are
exactly 0, hence so are the gradients of the padded weight rows, which start at 0
and stay 0 -- the padded network is numerically the reference's ragged one.

All parameters of all agents live in ONE flat fp32 buffer [N, P] (`ParamStore`):
row i holds agent i's parameters, every tensor is a strided view.  The gradient
and the RMSProp slot use the same layout, so clip + RMSProp is one fused kernel
and the data-parallel exchange is one RCCL all-reduce of the gradient buffer.
"""
import os
import warnings

import numpy as np
import torch

from .. import ops
from . import sequence

# parameters and their gradients are deliberate strided views of the flat [N,P] buffers (see ParamStore)
warnings.filterwarnings('ignore', message='grad and param do not obey the gradient layout contract')

F32 = torch.float32


def ortho_init(shape, scale=np.sqrt(2)):
    """agents/utils.py:10-23 -- draws from the GLOBAL np.random stream, like the reference."""
    a = np.random.standard_normal(shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == tuple(shape) else v
    return (scale * q.reshape(shape)).astype(np.float32)


class Layout:
    """Where agent i's (ragged) reference variable sits inside its padded tensor: `rows` / `cols` = padded indices
    in the reference's order (None = all), `exists` = the reference creates the variable at all, `fill` = value of
    the padded entries (0; -1e30 for the logit bias of actions an agent does not have)."""

    def __init__(self, rows=None, cols=None, exists=True, fill=0.0):
        self.rows, self.cols, self.exists, self.fill = rows, cols, exists, fill


def _layout(spec, i, shape):
    """Normalise a phase entry's 4th field: None (full tensor), a callable returning an int (valid leading rows:
    the homogeneous nets, where missing neighbours are trailing slots) or a Layout."""
    lay = spec(i) if callable(spec) else spec
    if lay is None:
        return Layout()
    if isinstance(lay, Layout):
        return lay
    return Layout(rows=np.arange(int(lay)))


class ParamStore:
    """Flat [N, P] parameter / gradient / RMSProp-slot buffers with named strided views.

    spec: list of phases; a phase is a list of (key, ref_name_fmt, padded_shape, layout_fn)
    created agent by agent -- the order in which the reference's tf.get_variable calls
    consume np.random (SURVEY.md 8a footnote "Init draw order")."""

    def __init__(self, n_agent, phases, device):
        self.N = n_agent
        self.phases = phases
        self.device = device
        self.index = {}
        off = 0
        ALIGN = 16            # floats: every tensor (and every agent row) starts on a 64-byte boundary,
        for phase in phases:  # as the float4 bias loads of the cell kernel and the GEMMs like it
            for key, fmt, shape, rows_fn in phase:
                size = int(np.prod(shape))
                self.index[key] = (off, size, tuple(shape), fmt, rows_fn)
                off += -(-size // ALIGN) * ALIGN
        self.P = off          # padding floats stay 0 (zero gradient), they never enter a norm or an export
        self.flat = torch.zeros(n_agent, self.P, dtype=F32, device=device)
        # the gradient and ONE more float behind it: the data-parallel exchange all-reduces `grad_wire` = [gradient | tail], the tail
        # carrying a rank's in-launch hand-off status to every other rank inside the same (single) collective (models.update_reduce)
        self.grad_wire = torch.zeros(n_agent * self.P + 1, dtype=F32, device=device)
        self.grad = self.grad_wire[:n_agent * self.P].view(n_agent, self.P)
        self.grad_tail = self.grad_wire[n_agent * self.P:]
        self.ms = torch.ones_like(self.flat)            # TF RMSProp slot starts at 1
        self.scratch = torch.zeros(n_agent, 64, dtype=F32, device=device)
        self.views = {}
        for key, (o, size, shape, _, _) in self.index.items():
            w = self.flat[:, o:o + size].view(n_agent, *shape)
            w.requires_grad_(True)
            w.grad = self.grad[:, o:o + size].view(n_agent, *shape)
            self.views[key] = w
        # trainable-entry mask: needed only when some padded entry could receive a gradient (a variable the reference
        # does not create for an agent); None for the homogeneous nets, whose padded rows see zero inputs
        self.mask = None
        if any(not _layout(fn, i, shape).exists for phase in phases for _, _, shape, fn in phase for i in range(n_agent)):
            host = np.zeros((n_agent, self.P), dtype=np.float32)
            for i, key, o, size, shape, lay in self._entries():
                if lay.exists:
                    blk = np.zeros(shape, dtype=np.float32)
                    self._region(blk, lay)[...] = 1.0
                    host[i, o:o + size] = blk.ravel()
            self.mask = torch.from_numpy(host).to(device)

    def __getitem__(self, key):
        return self.views[key]

    # ---- gradients of one update without an accumulation launch per parameter
    def begin_backward(self):
        """Instead of zeroing the flat gradient buffer: detach every parameter's `.grad`, so that autograd's AccumulateGrad takes
        the first incoming gradient tensor as it is (no `grad += g` launch per parameter); `end_backward` gathers them."""
        for w in self.views.values():
            w.grad = None

    def _zero_piece(self, width):
        z = self.__dict__.setdefault('_zero_pieces', {})
        if width not in z:
            z[width] = torch.zeros(self.N, width, dtype=F32, device=self.device)
        return z[width]

    def end_backward(self):
        """All parameter gradients -> the flat [N,P] buffer in ONE concatenation (alignment gaps and parameters without a
        gradient: zeros); `.grad` of every parameter is a view of the flat buffer again afterwards."""
        pieces, off = [], 0
        for key, (o, size, shape, _, _) in self.index.items():               # (in offset order)
            if o > off:
                pieces.append(self._zero_piece(o - off))
            g = self.views[key].grad
            pieces.append(self._zero_piece(size) if g is None else g.reshape(self.N, size))
            off = o + size
        if self.P > off:
            pieces.append(self._zero_piece(self.P - off))
        torch.cat(pieces, dim=1, out=self.grad)
        for key, (o, size, shape, _, _) in self.index.items():
            self.views[key].grad = self.grad[:, o:o + size].view(self.N, *shape)

    def _entries(self):
        """(agent, key, offset, size, padded shape, Layout) in the reference's variable creation order."""
        for phase in self.phases:
            for i in range(self.N):
                for key, fmt, shape, fn in phase:
                    o, size, _, _, _ = self.index[key]
                    yield i, key, o, size, tuple(shape), _layout(fn, i, shape)

    @staticmethod
    def _region(blk, lay):
        """View-like index of the reference-shaped part of a padded block (assignable through np.ix_)."""
        rows = np.arange(blk.shape[0]) if lay.rows is None else np.asarray(lay.rows)
        if blk.ndim == 1:
            return _Region(blk, (rows,))
        cols = np.arange(blk.shape[1]) if lay.cols is None else np.asarray(lay.cols)
        return _Region(blk, np.ix_(rows, cols))

    @staticmethod
    def _ref_shape(shape, lay):
        rows = shape[0] if lay.rows is None else len(lay.rows)
        if len(shape) == 1:
            return (rows,)
        return (rows, shape[1] if lay.cols is None else len(lay.cols))

    def _fmt_of(self, key):
        return self.index[key][3]

    def init_reference_order(self):
        """Initialise like the reference: weights `w*` = sqrt(2)-orthogonal drawn in
        variable-creation order from np.random, biases 0 (agents/utils.py:69-71, 95-99, 141-162)."""
        host = np.zeros((self.N, self.P), dtype=np.float32)
        for i, key, o, size, shape, lay in self._entries():
            blk = np.full(shape, lay.fill, dtype=np.float32) if lay.fill else np.zeros(shape, dtype=np.float32)
            if lay.exists:
                if key.endswith('_b'):
                    self._region(blk, lay)[...] = 0.0
                else:
                    self._region(blk, lay)[...] = ortho_init(self._ref_shape(shape, lay))
            host[i, o:o + size] = blk.ravel()
        with torch.no_grad():
            self.flat.copy_(torch.from_numpy(host))
        self.ms.fill_(1.0)
        self.grad.zero_()

    # ---- reference-named export / import (checkpoint + parity tests)
    def ref_variables(self):
        """[(reference variable name, np.ndarray with the reference's ragged shape)] in creation order."""
        host = self.flat.detach().cpu().numpy()
        out = []
        for i, key, o, size, shape, lay in self._entries():
            if lay.exists:
                blk = host[i, o:o + size].reshape(shape)
                out.append((self._fmt_of(key) % i, np.array(self._region(blk, lay).get())))
        return out

    def load_ref_variables(self, named):
        named = dict(named)
        host = self.flat.detach().cpu().numpy().copy()
        for i, key, o, size, shape, lay in self._entries():
            blk = np.full(shape, lay.fill, dtype=np.float32) if lay.fill else np.zeros(shape, dtype=np.float32)
            if lay.exists:
                self._region(blk, lay)[...] = np.asarray(named[self._fmt_of(key) % i]).reshape(self._ref_shape(shape, lay))
            host[i, o:o + size] = blk.ravel()
        with torch.no_grad():
            self.flat.copy_(torch.from_numpy(host))


class _Region:
    """blk[idx] with assignment and read through one object (np.ix_ fancy indices are not views)."""

    def __init__(self, blk, idx):
        self.blk, self.idx = blk, idx

    def __setitem__(self, _, value):
        self.blk[self.idx] = value

    def get(self):
        return self.blk[self.idx]


class BatchedPolicy:
    """Common part: heads, rollout step (Q1 double step handled by the caller), unroll."""

    name = 'policy'
    fused_coupled = True          # coupled policies: use agents/sequence.py in the update

    def __init__(self, n_feat, n_a, neighbor_mask, n_fc=64, n_h=64, device='cuda', n_feat_ls=None, n_a_ls=None, obs_order=None):
        """n_feat / n_a: own observation width / action count of an agent -- the maxima when agents differ.
        obs_order (IA2C family only): per agent the neighbours in the order the ENV concatenates them into that agent's
        observation (the ATSC envs list them north-east-south-west, atsc_env.py:263-271; None: ascending index, CACC).  The
        engine's slab and its gather kernels stay in ascending-slot order; the order only decides which rows of the padded
        observation / fingerprint weights the reference's variable rows occupy (`_L_slots`, `_L_fp_obs`) and how the E = 1
        API scatters an observation vector into the slab (models._obs_to_slab) -- so initial draws, checkpoints and the
        function computed from the env's vectors are the reference's.
        n_feat_ls / n_a_ls: the per-agent values of heterogeneous systems (`identical=False` nets of the reference:
        agents/utils.py:220-341, 420-512, 602-719; policies.py:59-77 with na_dim_ls).  Those run as the SAME padded
        batched network: observations, fingerprints, logits and one-hots are padded to the maxima, the reference's
        ragged variables occupy the matching rows / columns of the padded tensors (ParamStore Layouts), actions an agent
        does not have carry a -1e30 logit bias (probability exactly 0: never drawn, no entropy, no gradient), and
        entries of variables the reference does not create (fingerprint / message layers of an agent without
        neighbours) are masked out of the update."""
        self.device = torch.device(device)
        self.nbr_idx, self.nbr_cnt = ops.neighbor_table(neighbor_mask, self.device)
        self.N = len(self.nbr_cnt)
        self.m_max = self.nbr_idx.shape[1]
        self.n_feat, self.n_a, self.n_fc, self.n_h = n_feat, n_a, n_fc, n_h
        self.n_own = [int(x) for x in n_feat_ls] if n_feat_ls is not None else [n_feat] * self.N
        self.n_a_ls = [int(x) for x in n_a_ls] if n_a_ls is not None else [n_a] * self.N
        # the reference's criterion (models.py:89-96): agents are "identical" iff their action counts agree
        self.hetero = max(self.n_a_ls) != min(self.n_a_ls)
        if self.hetero and (max(self.n_own) != n_feat or max(self.n_a_ls) != n_a):
            raise ValueError('heterogeneous agents: n_feat / n_a must be the maxima of n_feat_ls / n_a_ls')
        tab = self.nbr_idx.cpu().numpy()
        self.nbrs = [[int(j) for j in tab[i] if j >= 0] for i in range(self.N)]
        self.obs_order = self.nbrs if obs_order is None else [[int(j) for j in o] for o in obs_order]
        if any(sorted(o) != n for o, n in zip(self.obs_order, self.nbrs)):
            raise ValueError('obs_order must list exactly the neighbours of every agent')
        self.obs_permuted = self.obs_order != self.nbrs
        # [own | neighbours] gather table for compact observations: slot 0 = the agent itself
        self.nbr_self = torch.cat([torch.arange(self.N, dtype=torch.int32, device=self.device).view(-1, 1), self.nbr_idx], dim=1)
        self.n_obs = n_feat * (1 + self.m_max)          # gathered observation slab width
        self.n_na = n_a * self.m_max                    # neighbour one-hot width
        self.params = ParamStore(self.N, self._phases(), self.device)

    # -- helpers for the ragged reference shapes
    def _m(self, i):
        return self.nbr_cnt[i]

    def _L_slots(self, i):
        """Rows of a weight over the gathered observation slab [own | nbr_1 | ... ] (slots n_feat wide)."""
        F = self.n_feat
        if not self.hetero and not self.obs_permuted:
            return F * (1 + self._m(i))
        rows = list(range(self.n_own[i]))
        for j in self.obs_order[i]:                      # the reference variable's row blocks follow the env's order;
            k = self.nbrs[i].index(j)                    # neighbour j sits in (ascending) slot k + 1 of the slab
            rows += [(k + 1) * F + f for f in range(self.n_own[j])]
        return Layout(rows=rows)

    def _L_fp(self, i, order=None):
        """Rows of a weight over the gathered neighbour fingerprints / action one-hots (slots n_a wide; the critic's one-hots
        are in mask order, i.e. ascending)."""
        A = self.n_a
        if not self.hetero and order is None:
            return A * self._m(i)
        order = self.nbrs[i] if order is None else order
        return Layout(rows=[self.nbrs[i].index(j) * A + a for j in order for a in range(self.n_a_ls[j])],
                      exists=(not self.hetero) or self._m(i) > 0)

    def _L_fp_obs(self, i):
        """Rows of a weight over the fingerprints that arrive INSIDE the observation (IA2C-FP: the env's neighbour order)."""
        return self._L_fp(i, order=self.obs_order[i] if self.obs_permuted else None)

    def _L_nbr(self, rows_per_nbr=None):
        """A variable of the neighbour-facing layers: `rows_per_nbr` leading rows per neighbour (None: full tensor);
        in heterogeneous nets it exists only for agents that have neighbours."""
        def f(i):
            m = self._m(i)
            if not self.hetero:
                return None if rows_per_nbr is None else rows_per_nbr * m
            return Layout(rows=None if rows_per_nbr is None else list(range(rows_per_nbr * m)), exists=m > 0)
        return f

    def _L_blocks(self, width, n_blocks):
        """LSTM input weight over `n_blocks` concatenated encodings: an agent without neighbours has the first only."""
        def f(i):
            if not self.hetero or self._m(i) > 0:
                return None
            return Layout(rows=list(range(width)))
        return f

    def _head_phase(self, pi_fmt, v_fmt):
        H, A = self.n_h, self.n_a
        if not self.hetero:
            return [('pi_w', pi_fmt + '/w', (H, A), None), ('pi_b', pi_fmt + '/b', (A,), None),
                    ('v_w', v_fmt + '/w', (H + self.n_na, 1), lambda i: H + A * self._m(i)),
                    ('v_b', v_fmt + '/b', (1,), None)]
        return [('pi_w', pi_fmt + '/w', (H, A), lambda i: Layout(cols=list(range(self.n_a_ls[i])))),
                ('pi_b', pi_fmt + '/b', (A,), lambda i: Layout(rows=list(range(self.n_a_ls[i])), fill=-1e30)),
                ('v_w', v_fmt + '/w', (H + self.n_na, 1),
                 lambda i: Layout(rows=list(range(H)) + [H + r for r in self._L_fp(i).rows])),
                ('v_b', v_fmt + '/b', (1,), None)]

    def _scratch(self, name, h):
        """Per-(name, row count) scratch [N,E,H].  Buffers are never dropped or replaced: a captured hipGraph holds the
        pointer of the one its rollout used, and an evaluation with a different row count (BatchedTrainer.evaluate) must
        not free it under the graph's feet."""
        pool = self.__dict__.setdefault('_scratch_pool', {})
        key = (name, h.shape[1], h.device)
        if key not in pool:
            pool[key] = torch.empty(self.N, h.shape[1], self.n_h, dtype=F32, device=h.device)
        return pool[key]

    # -- heads: policies.py:50-77
    def pi(self, h):
        p = self.params
        return torch.softmax(torch.baddbmm(p['pi_b'].unsqueeze(1), h, p['pi_w']), dim=-1)

    def heads(self, h, action):
        """Actor and critic heads of the update for all rows: h [N,rows,H], action [rows,N] u8 (env-major bytes; the
        critic's neighbour one-hots are gathered from them) -> (logits [N,rows,A], v [N,rows]).  H = 64: one skinny
        GEMM forward, one streaming pass backward (ops.heads); otherwise plain batched GEMMs."""
        p = self.params
        H, A = self.n_h, self.n_a
        if ops.heads_supported(h, A, self.nbr_idx):
            return ops.heads(h, p['pi_w'], p['pi_b'], p['v_w'], p['v_b'], action, self.nbr_idx, A)
        w = torch.cat([p['pi_w'], p['v_w'][:, :H]], dim=2)
        b = torch.cat([p['pi_b'], p['v_b']], dim=1)
        out = torch.baddbmm(b.unsqueeze(1), h, w)
        na = ops.nbr_onehot(action, self.nbr_idx, A)
        return out[..., :A], out[..., A] + torch.bmm(na, p['v_w'][:, H:]).squeeze(-1)

    def value(self, h, na_onehot, out=None):
        """v = [h, onehot(neighbour actions)] @ Wv + b (policies.py:59-77), without the concat;
        `out` [N,rows] (no-grad rollout) receives the result in place."""
        p = self.params
        H = self.n_h
        v = torch.baddbmm(p['v_b'].unsqueeze(1), h, p['v_w'][:, :H])
        if out is not None:
            torch.baddbmm(v, na_onehot, p['v_w'][:, H:], out=out.unsqueeze(-1))
            return out
        return torch.baddbmm(v, na_onehot, p['v_w'][:, H:]).squeeze(-1)

    # -- rollout (no autograd): encode once per lock-step, then one recurrent step per call
    def encode(self, x, fp_prev, out=None):
        """The h-independent part of the LSTM input for one lock-step: x [E,N,n_obs] env-major slab
        (any view with that shape), fp_prev [N,E,A] previous-step policies.  Both the policy step and
        the value re-step (quirk Q1) of a lock-step share it.  `out` (x-side mode only): the slot of the saved
        activations [N,E,KX] the LSTM input is written to."""
        with torch.no_grad():
            if out is not None:
                return self._enc_infer(x.transpose(0, 1), fp_prev, out=out)
            return self._enc_infer(x.transpose(0, 1), fp_prev)

    def step(self, enc, h, c, done, h_out, c_out, done_is_zero=False, second=False):
        """One LSTM step from (h, c) [N,E,H] with done [E] f32 -> writes (h', c') into (h_out, c_out),
        which MAY alias (h, c).  H = 64: ONE fused MFMA kernel computes (h*(1-done)) @ Wh on top of the
        x-side addends and applies the cell in its epilogue (csrc/lstm_mfma.hip).  Other widths: a plain
        batched GEMM + the cell kernel (which takes the x-side part as a second addend: no copy GEMM)."""
        with torch.no_grad():
            z1, z2, xs = self._recur_addends(enc, h, second=second)
            wh, b = self.params[self.k_wh], self.params[self.k_b]
            if self.n_h == ops.FUSED_H:
                ops.lstm_step_fused(h, wh, b, z1, z2, c, done, None, c_out, h_out, xs=xs)
            else:
                hk = h if done_is_zero else h * (1.0 - done).view(1, -1, 1)
                z = torch.bmm(hk, wh) if z2 is None else torch.baddbmm(z2, hk, wh)
                ops.lstm_cell_infer(z, b, c, done, c_out, h_out, z2=z1)
        return h_out, c_out

    @property
    def fused_heads(self):
        """The actor / critic heads can ride in the fused step's epilogue (csrc/lstm_mfma.hip)."""
        return self.n_h == ops.FUSED_H and self.n_a <= ops.HEAD_MAX_A

    def step_policy(self, enc, h, c, done, h_out, c_out, pi_out, act_out, done_is_zero=False, gates=None, save=None,
                    **draw):
        """forward('p') + the action draw of one lock-step (policies.py:119-123, utils.py:135-141): one LSTM step,
        pi -> pi_out [N,E,A], actions -> act_out [E,N] u8.  `draw`: sample_actions' mode / u / seed / env_id_base /
        step / step_dev.  One kernel when `fused_heads`."""
        with torch.no_grad():
            if self.fused_heads:
                # (in place, another agent's block could overwrite h while this one still gathers it for its message term)
                z1, z2, xs = self._recur_addends(enc, h, save=save, fuse_msg=self.msg_inplace_ok or h_out.data_ptr() != h.data_ptr())
                p = self.params
                ops.lstm_step_policy(h, p[self.k_wh], p[self.k_b], z1, z2, c, done, c_out, h_out, p['pi_w'], p['pi_b'],
                                     pi_out, act_out, xs=xs, gates=gates, **draw)
            else:
                self.step(enc, h, c, done, h_out, c_out, done_is_zero)
                pi_out.copy_(self.pi(h_out))
                ops.sample_actions(pi_out, act_out, **draw)
        return pi_out, act_out

    @property
    def fused_pv(self):
        """forward('p') and forward('v') of a lock-step fit ONE kernel: heads fuse and the recurrence has no
        cross-agent term (the value re-step of a coupled net needs the other agents' new h)."""
        return self.fused_heads and not self.coupled

    @property
    def fused_pv_coupled(self):
        """forward('p') and forward('v') of a lock-step in ONE kernel for a COUPLED net: the message term is computed inside
        the step kernel and the blocks hand their new h over inside the launch (ops.step_handoff_supported).  Decided per
        call site (the number of replicas matters): `pv_one_launch(E)`."""
        return self.coupled and self.fused_heads and self.msg_kind != ops.MSG_DIAL and self._msg() is not None

    def pv_one_launch(self, E):
        if self.fused_pv:
            return True
        K = self.n_h * self.m_max if self.msg_kind == ops.MSG_GATHER_RELU else self.n_h
        return self.fused_pv_coupled and ops.step_handoff_supported(self.N, E, self.device, K=K)

    def _sync_words(self, E):
        """Flag words of the one-launch step per replica count, never dropped or replaced (a captured hipGraph holds the
        pointer of the one its rollout used, cf. `_scratch`)."""
        pool = self.__dict__.setdefault('_sync_pool', {})
        if E not in pool:
            pool[E] = ops.step_sync_words(self.N, E, self.device)
        return pool[E]

    def encodes_in_step(self, E, compact):
        """The one-launch step of this net also runs the observation encoder (no separate encoder launch per lock-step)."""
        return False

    _bits_steps = 0          # lock-steps of the running batch whose encoder sign image the rollout wrote (models._relu_bits)

    def invalidate_cached_msg(self):
        """(lstm_dial keeps message vectors of the last policy step: see DIALMultiAgentPolicy)"""

    def enc_in_kernel(self, E, compact):
        """Uncoupled nets: the policy + value launch of a lock-step also runs BOTH input encoders (csrc/lstm_mfma.hip ENC; the
        caller hands `step_policy_value` the env's compact observation and the fingerprints as `ob`)."""
        return False

    @property
    def env_step_in_kernel(self):
        """With the encoders inside the lock-step launch (`enc_in_kernel`), should the batched CACC engine put the env step there
        too (ONE launch per lock-step)?  A measured choice per net: profiles/r05_ab_lockstep.txt, r06_ab_lockstep_nc.txt."""
        return True

    def step_policy_value(self, enc, h, c, done, pi_out, act_out, v_out, h_out=None, c_out=None, gates=None,
                          defer_action_term=False, save=None, ob=None, carry=None, **draw):
        """Both halves of a lock-step decision (Trainer._get_policy + _get_value, utils.py:129-149) in one kernel:
        advances (h, c) by the policy step -- in place, or into (h_out, c_out) with the gates saved for the update --;
        the value comes from the re-stepped copy (quirk Q1).  Coupled nets (`pv_one_launch`): never in place; `save`
        as in step_policy (the POLICY step's message term is what the update needs).  ob (`encodes_in_step` nets): the env's
        compact observation [E,N,F] of this lock-step -- `enc` is then the slot the kernel WRITES the encoder's output to.
        carry (coupled nets, one launch): dict(carry_in, carry_out[, mean_next]) -- the re-step's message term is the next lock-step's
        policy-step message term (the neighbours' un-masked new h, Q3); see include/nmarl.h nmarl_msg_t."""
        with torch.no_grad():
            if self.coupled:
                z1, z2, xs = self._recur_addends(enc, h, save=save, fuse_msg=True)
                xs[4]['sync'] = self._sync_words(h.shape[1])
                if carry:
                    xs[4].update(carry)
                if isinstance(ob, dict) and 'genv' in ob:      # lstm_ic3 on the grid: observation encoder AND env step inside the launch
                    xs[4]['ob'], xs[4]['genv'] = self._ob_spec(ob['x']), ob['genv']
                elif isinstance(ob, dict):       # lstm_comm: both input encoders (and the env step) inside the launch, see `enc_in_kernel`
                    xs[4]['enc_spec'] = self._enc_spec(ob['x'], ob['fp'], None, ob.get('env'), ob.get('bits'))
                elif ob is not None:
                    xs[4]['ob'] = self._ob_spec(ob)
            elif ob is not None:
                # the encoders run inside the launch: ob = dict(x = compact observation [E,N,5], fp = previous policies [N,E,4]);
                # `enc` = where their output (the LSTM input) is kept for the update, or None
                z1, z2, xs = None, None, (self._enc_spec(ob['x'], ob['fp'], enc, ob.get('env'), ob.get('bits')), self.params[self.k_wx], self._img)
            else:
                z1, z2, xs = self._recur_addends(enc, h)
            p = self.params
            ops.lstm_step_policy_value(h, p[self.k_wh], p[self.k_b], z1, z2, c, done, p['pi_w'], p['pi_b'], pi_out, act_out,
                                       p['v_w'], p['v_b'], self.nbr_idx, self.n_a, v_out, xs=xs, h_out=h_out, c_out=c_out,
                                       gates=gates, defer_action_term=defer_action_term, **draw)
        return act_out

    @property
    def bptt_takes_head_dy(self):
        """The backward of `unroll_saved` understands ops.head_dy_placeholder (the heads' dL/dh as dy8, expanded inside the
        BPTT kernel): the uncoupled nets' recurrence (ops._lstm_seq_x_backward) and the coupled one (agents/sequence.py; DIAL's
        step-wise recurrence turns it into the tensor first)."""
        if self.coupled and os.environ.get('NMARL_BPTT_HEAD_DY_COUPLED', '1') == '0':      # (A/B switch for the coupled kernels alone)
            return False
        return self.xside

    @property
    def can_save_acts(self):
        """The rollout can hand its activations (LSTM inputs, gates, state sequences, message terms) to the update,
        which then needs no forward pass: nets whose recurrent step is the fused x-side kernel."""
        return self.xside and self.fused_heads

    def unroll_saved(self, X, FP, S, G, Hall, Call, done, masked_steps=None, S_ext=None, S_bits=None):
        """`unroll` for a batch whose forward pass the rollout already did with the CURRENT weights: S [N,T,E,KX] the
        LSTM inputs, G the gates, Hall / Call [N,T+1,E,H] the state sequences it saved.  Sets up the backward only.
        S_ext: the [N,T+1,E,KX] buffer S is the first T slabs of (last slab zero), see ops._lstm_seq_x_backward."""
        T, E = done.shape
        Xv = X.reshape(T * E, self.N, X.shape[-1]).transpose(0, 1)          # gathered [.., n_obs] or compact [.., n_feat] slab
        if self.coupled:
            return self._unroll_saved_coupled(Xv, FP, S, G, Hall, Call, done, masked_steps, S_ext, S_bits)
        if S_bits is not None:           # the sign image of S the lock-step kernel's encoders wrote (FPPolicy): S is not re-read
            s = self._enc(Xv, FP, saved=S.view(self.N, T * E, S.shape[-1]), bits=S_bits.view(self.N, T * E, 4))
        else:
            s = self._enc(Xv, FP, saved=S.view(self.N, T * E, S.shape[-1]))
        Hs = ops.lstm_sequence_saved(s.view(self.N, T, E, s.shape[-1]), self.params[self.k_wx], self.params[self.k_wh],
                                     self.params[self.k_b], G, Hall, Call, done, masked_steps, s_ext=S_ext)
        return Hs.reshape(self.N, T * E, self.n_h)

    def save_spec(self):
        """Extra per-step tensors a coupled net saves besides S / G / Hall / Call: {name: width}."""
        return {}

    def _unroll_saved_coupled(self, Xv, FP, S, G, Hall, Call, done, masked_steps, S_ext=None, S_bits=None):
        T, E = done.shape
        kind, wx, w_msg, b_msg, mfc_w, mfc_b = self._seq_args()
        # (S_bits: the sign image of the encoders' output, written by the lock-step kernel that ran them -- lstm_comm's one-launch form)
        enc = self._enc_saved(Xv, FP, S) if S_bits is None else self._enc_saved(Xv, FP, S, bits=S_bits.view(self.N, T * E, 4))
        extra = dict(getattr(self, '_extra', {}))
        if kind == 'dial' and 'A2' in getattr(self, '_extra_full', {}) and extra.get('A2') is not None:
            extra['A2'] = self._extra_full['A2']          # the (T + 1)-slab buffer itself: the weight gradient reads it in place
        if kind == 'ic3':
            # the rollout's mean_nbr(h_{t-1}) rows ((T + 1)-slab buffer, last slab zero) if every policy step of this batch kept them
            mm = getattr(self, '_extra_full', {}).get('MM') if getattr(self, '_mm_was_saved', False) else None
            extra['A1'] = mm if (mm is not None and mm.shape[1] == T + 1) else None
        Hs = sequence.coupled_sequence_saved(kind, self.nbr_idx, masked_steps, enc.view(self.N, T, E, enc.shape[-1]), done,
                                             self.params[self.k_wx], self.params[self.k_wh], self.params[self.k_b],
                                             w_msg, b_msg, mfc_w, mfc_b, G, Hall, Call, S, extra, s_ext=S_ext)
        return Hs.reshape(self.N, T * E, self.n_h)

    def step_value(self, enc, h, c, done, h_out, c_out, action, v_out, done_is_zero=False):
        """forward('v') of one lock-step (policies.py:124-133): the LSTM re-step and the critic on
        [h', onehot(neighbours' actions)], actions given as the env-major byte array action [E,N] -> v_out [N,E]."""
        with torch.no_grad():
            if self.fused_heads:
                z1, z2, xs = self._recur_addends(enc, h, second=True, fuse_msg=True)
                p = self.params
                ops.lstm_step_value(h, p[self.k_wh], p[self.k_b], z1, z2, c, done, c_out, h_out, p['v_w'], p['v_b'],
                                    action, self.nbr_idx, self.n_a, v_out, xs=xs)
            else:
                self.step(enc, h, c, done, h_out, c_out, done_is_zero, second=True)
                self.value(h_out, ops.nbr_onehot(action, self.nbr_idx, self.n_a), out=v_out)
        return v_out

    def fused_env_encode(self, fp_next, out):
        """Arguments for the env kernel to run THIS policy's input encoders behind its step (envs/cacc_env.py `encode`), or
        None if the encoders do not have that shape: relu layers of 64 outputs over the compact CACC observation
        ([own | 2 neighbours] x 5 features) and, optionally, the 2 neighbours' 4-wide fingerprints."""
        return None

    def _fused_spec(self, ob, fp_keys, fp_next, out):
        """ob = (weight key, bias key) of the observation layer, fp_keys likewise for the fingerprint layer or None."""
        p = self.params
        if self.hetero or self.n_feat != 5 or self.m_max != 2 or self.n_a != 4 or p[ob[0]].shape[1:] != (15, 64):
            return None
        d = dict(w_ob=p[ob[0]], b_ob=p[ob[1]], nbr_idx=self.nbr_idx, out=out, act=ops.BIAS_RELU)
        if fp_keys is not None:
            if p[fp_keys[0]].shape[1:] != (8, 64):
                return None
            d.update(w_fp=p[fp_keys[0]], b_fp=p[fp_keys[1]], fp=fp_next)
        return d

    def _enc_one_launch(self, xv, width):
        """Observation and fingerprint encoders fit the multi-layer fc kernel (inputs <= 64 wide, 64 outputs)."""
        return width == ops.FC_J and self.n_obs <= ops.FC_MAX_F and self.n_na <= ops.FC_MAX_F

    def _compact(self, xv):
        """xv holds each agent's OWN features only ([N,rows,n_feat], the env's compact observation): the observation
        encoder gathers [own | neighbours] through `nbr_self` inside the fc kernel."""
        return xv.shape[2] == self.n_feat and self.n_obs != self.n_feat

    def _ob_part(self, xv, w_key, b_key):
        p = self.params
        return (xv, p[w_key], p[b_key], self.nbr_self if self._compact(xv) else None)

    def _fc_ob_infer(self, xv, w_key, b_key, act, out=None):
        """The observation encoder layer act([x_i, x_nbr] W + b) (policies.py:145, 177; agents/utils.py:186-197, 395-399)
        on the gathered slab or on the compact observation."""
        if self._compact(xv):
            return ops.fc_fwd_multi([self._ob_part(xv, w_key, b_key)], act, out=out)
        return self._fc_infer(xv, w_key, b_key, act, out=out)

    def expand_obs(self, X):
        """Compact observations [..., N, n_feat] -> the gathered slab [..., N, n_obs] ([own | neighbours ascending],
        zero padded): what the update's encoders read (one gather pass over the batch)."""
        idx = self.nbr_self.long()
        g = X[..., idx.clamp(min=0), :] * (idx >= 0).to(X.dtype).unsqueeze(-1)       # [..., N, 1+m, n_feat]
        return g.reshape(*X.shape[:-1], self.n_obs)

    def _fc_infer(self, x, w_key, b_key, act, out=None):
        """act(x @ W + b) with the bias/activation fused in one pass (no autograd); `out` may be a column
        block of a wider buffer, which concatenates partial encodings without a copy."""
        w, b = self.params[w_key], self.params[b_key]
        if ops.fc_supported(x, w):
            return ops.fc_fwd(x, w, b, act, out=out)          # small-K layer: one streaming kernel (csrc/fc.hip)
        return ops.bias_act_(torch.bmm(x, w), b, act, out=out)

    def _recur_addends(self, enc, h, second=False, save=None, fuse_msg=False):
        """(zadd1, zadd2, xs): everything of the LSTM pre-activation except (h*(1-done)) @ Wh and the bias -- as
        ready-made addends [N,E,4H] and / or as xs = (x, wx, weight image[, x2]): an input [x | x2] [N,E,KX] whose
        product with wx the fused step computes itself (ops.lstm_step_fused).
        Coupled nets compute their message terms from h here.  second: the call is the value re-step of a lock-step
        (quirk Q1) -- what the policy step wrote into `enc` / the save slots must survive.  save: slots of the saved
        activations (dict of [N,E,*] tensors) the policy step's message terms are written to.  fuse_msg: the caller is a
        head step (policy / value), whose kernel can compute the message term itself (`_msg`)."""
        if self.xside:
            return None, None, (enc, self.params[self.k_wx], self._img)
        return enc, None, None

    # -- x-side product inside the fused step (uncoupled nets: the LSTM input is the encoders' output itself)
    k_wx = None
    _img = None

    @property
    def xside(self):
        """The step kernel multiplies the LSTM input by Wx itself (csrc/lstm_mfma.hip lstm_step_x_kernel): no
        [rows,4H] pre-activation tensor, no separate GEMM.  Needs H = 64 and an input width that is a multiple of 32."""
        return self.k_wx is not None and ops.xside_supported(self.params[self.k_wx].shape[1], self.n_h)

    def refresh_wimage(self):
        """Rebuild the chunked [Wx; Wh] image the x-side step reads; call after every change of the weights (the
        batched engine does it at the first lock-step of a batch and before the update's forward pass)."""
        if self.xside:
            self._img = ops.lstm_wimage(self.params[self.k_wx], self.params[self.k_wh], out=self._img)
            if self.msg_kind and ops.msg_supported(self.msg_kind, self.m_max, self.n_h):
                self._msg_img = ops.lstm_msg_wimage(self.params['w_msg'], out=self._msg_img)

    msg_kind = 0            # ops.MSG_*: the message term the step kernel can compute itself (coupled nets)
    msg_inplace_ok = False  # the in-kernel message term reads no h of other agents (lstm_dial: the senders' message vectors)
    _msg_img = None

    def _msg(self, **kw):
        """The in-kernel message term of a policy / value step (ops._step_x `msg`), or None if it does not fit."""
        if not (self.xside and self.msg_kind and self._msg_img is not None):
            return None
        p = self.params
        return dict(kind=self.msg_kind, nbr_idx=self.nbr_idx, w_msg=p['w_msg'], b_msg=p['w_msg_b'], img=self._msg_img, **kw)

    # -- n_step unroll for the update (autograd)
    def unroll(self, X, FP, done, h0, c0, masked_steps=None):
        """X [T,E,N,n_obs] env-major, FP [N,T*E,A] previous-step policies, done [T,E] f32
        (pre-step), (h0, c0) [N,E,H] -> Hs [N,T*E,H]."""
        T, E = done.shape
        Xv = X.reshape(T * E, self.N, X.shape[-1]).transpose(0, 1)       # [N, T*E, n_obs] (or compact: n_feat), no copy
        enc = self._enc(Xv, FP)
        if not self.coupled:
            # no cross-agent term inside the recurrence: fused sequence op (one wgrad GEMM, one bias
            # reduction, no per-step autograd nodes)
            if self.xside:      # enc = the LSTM input s; s @ Wx happens inside the step kernel
                self.refresh_wimage()
                Hs = ops.lstm_sequence_x(enc.view(self.N, T, E, enc.shape[-1]), self.params[self.k_wx],
                                         self.params[self.k_wh], self.params[self.k_b], h0, c0, done, masked_steps,
                                         self._img)
            else:
                Hs = ops.lstm_sequence(enc.view(self.N, T, E, enc.shape[-1]), self.params[self.k_wh],
                                       self.params[self.k_b], h0, c0, done, masked_steps)
            return Hs.reshape(self.N, T * E, self.n_h)
        if self.fused_coupled:
            # cross-agent recurrences: manual BPTT in one autograd node (agents/sequence.py)
            kind, wx, w_msg, b_msg, mfc_w, mfc_b = self._seq_args()
            Hs = sequence.coupled_sequence(kind, self.nbr_idx, masked_steps, enc.view(self.N, T, E, enc.shape[-1]), h0, c0,
                                           done, wx, self.params[self.k_wh], self.params[self.k_b], w_msg, b_msg,
                                           mfc_w, mfc_b)
            return Hs.reshape(self.N, T * E, self.n_h)
        # reference path (kept for cross-checking the fused one): per-step autograd nodes.  Per-step views
        # via ONE unbind: its backward is a single stack, whereas slicing `enc` inside the loop would make
        # autograd materialise and add T full-size zero tensors (O(T^2) traffic)
        enc_steps = enc.view(self.N, T, E, enc.shape[-1]).unbind(1)
        h, c = h0, c0
        hs = []
        wh, b = self.params[self.k_wh], self.params[self.k_b]
        for t in range(T):
            zx = self._recur_in(enc_steps[t], h)
            keep = (1.0 - done[t]).view(1, -1, 1)
            z = torch.baddbmm(zx, h * keep, wh)
            h, c = ops.lstm_cell(z, b, c, done[t])
            hs.append(h)
        return torch.stack(hs, dim=1).reshape(self.N, T * E, self.n_h)


class LstmPolicy(BatchedPolicy):
    """IA2C: fc(n_s -> n_fc, relu) -> LSTM -> heads (policies.py:136-149)."""
    name = 'lstm'
    k_wh, k_b, k_wx = 'lstm_wh', 'lstm_b', 'lstm_wx'
    k_ob = 'fc_w'                 # the observation encoder's weight
    coupled = False               # the recurrence has no cross-agent term -> fused sequence op

    def _phases(self):
        nf, H, F = self.n_fc, self.n_h, self.n_feat
        return [[('fc_w', 'lstm_%d/fc/w', (self.n_obs, nf), self._L_slots),
                 ('fc_b', 'lstm_%d/fc/b', (nf,), None),
                 ('lstm_wx', 'lstm_%d/lstm/wx', (nf, 4 * H), None),
                 ('lstm_wh', 'lstm_%d/lstm/wh', (H, 4 * H), None),
                 ('lstm_b', 'lstm_%d/lstm/b', (4 * H,), None)] + self._head_phase('lstm_%d/pi', 'lstm_%d/v')]

    def _enc(self, xv, fp, saved=None):
        """The LSTM input s (x-side mode), else the x-side pre-activation s @ Wx [N,rows,4H] (bias is added in the
        cell kernel).  saved: s as the rollout computed it (x-side mode) -- only the backward is set up."""
        p = self.params
        s = ops.fc_concat([self._ob_part(xv, 'fc_w', 'fc_b')], ops.BIAS_RELU, saved=saved)
        return s if self.xside else ops.linear(s, p['lstm_wx'])

    def _enc_infer(self, xv, fp, out=None):
        s = self._fc_ob_infer(xv, 'fc_w', 'fc_b', ops.BIAS_RELU, out=out)
        return s if self.xside else torch.bmm(s, self.params['lstm_wx'])

    def fused_env_encode(self, fp_next, out):
        return self._fused_spec(('fc_w', 'fc_b'), None, fp_next, out) if self.xside else None

    enc_writes_bits = False       # (the 16-byte sign image belongs to the two-layer encodings: FPPolicy, NCMultiAgentPolicy)

    def enc_in_kernel(self, E, compact):
        """IA2C / ConseNet on CACC (round 6): the policy + value launch runs the observation encoder itself (csrc/lstm_mfma.hip ENC 2)."""
        m_enc = self.m_max if self.params['fc_w'].shape[1] == self.n_obs else 0       # ConseNet: own features only
        return bool(compact) and self.xside and self.fused_pv and not self.hetero and \
            ops.step_enc1_supported(self.n_feat, m_enc, self.n_fc, self.n_h, self.N)

    def _enc_spec(self, x, fp, out, env=None, bits=None):
        p = self.params
        return ops.step_enc_spec(x, fp, p['fc_w'], p['fc_b'], None, None, self.nbrs, out=out, env=env, bits=None)

    def _recur_in(self, enc, h):
        return enc


class FPPolicy(LstmPolicy):
    """IA2C_FP: fcs(obs) || fcp(neighbour fingerprints) -> LSTM(2 n_fc) (policies.py:163-185)."""
    k_ob = 'fcs_w'
    enc_writes_bits = True

    def _phases(self):
        nf, H, F, A = self.n_fc, self.n_h, self.n_feat, self.n_a
        return [[('fcs_w', 'lstm_%d/fcs/w', (self.n_obs, nf), self._L_slots),
                 ('fcs_b', 'lstm_%d/fcs/b', (nf,), None),
                 ('fcp_w', 'lstm_%d/fcp/w', (self.n_na, nf), self._L_fp_obs),
                 ('fcp_b', 'lstm_%d/fcp/b', (nf,), self._L_nbr()),
                 ('lstm_wx', 'lstm_%d/lstm/wx', (2 * nf, 4 * H), self._L_blocks(nf, 2)),
                 ('lstm_wh', 'lstm_%d/lstm/wh', (H, 4 * H), None),
                 ('lstm_b', 'lstm_%d/lstm/b', (4 * H,), None)] + self._head_phase('lstm_%d/pi', 'lstm_%d/v')]

    def _enc(self, xv, fp, saved=None, bits=None):
        p = self.params
        nf = self.n_fc
        # tf.concat([hx, hp]) @ wx as ONE K = 2 nf GEMM; both layers write their block of the concatenation in place; the
        # neighbour gathers of the (compact) observation and of the fingerprints happen inside the fc kernels
        s = ops.fc_concat([self._ob_part(xv, 'fcs_w', 'fcs_b'), (fp, p['fcp_w'], p['fcp_b'], self.nbr_idx)], ops.BIAS_RELU, saved=saved,
                          bits=bits)
        return s if self.xside else ops.linear(s, p['lstm_wx'])

    def _enc_infer(self, xv, fp, out=None):
        p = self.params
        nf = self.n_fc
        if self._enc_one_launch(xv, nf):
            # [hx | hp] (policies.py:181) by ONE kernel: both layers, the fingerprint gather folded into the second
            s = ops.fc_fwd_multi([self._ob_part(xv, 'fcs_w', 'fcs_b'), (fp, p['fcp_w'], p['fcp_b'], self.nbr_idx)],
                                 ops.BIAS_RELU, out=out)
        else:
            s = torch.empty(self.N, xv.shape[1], 2 * nf, dtype=F32, device=xv.device) if out is None else out
            self._fc_ob_infer(xv, 'fcs_w', 'fcs_b', ops.BIAS_RELU, out=s[:, :, :nf])
            self._fc_infer(ops.nbr_gather(fp, self.nbr_idx), 'fcp_w', 'fcp_b', ops.BIAS_RELU, out=s[:, :, nf:])
        return s if self.xside else torch.bmm(s, p['lstm_wx'])                          # else ONE K = 2 nf GEMM

    def fused_env_encode(self, fp_next, out):
        return self._fused_spec(('fcs_w', 'fcs_b'), ('fcp_w', 'fcp_b'), fp_next, out) if self.xside else None

    def enc_in_kernel(self, E, compact):
        return bool(compact) and self.xside and self.fused_pv and not self.hetero and \
            ops.step_enc_supported(self.n_feat, self.n_a, self.m_max, self.n_fc, self.n_h, self.N)

    def _enc_spec(self, x, fp, out, env=None, bits=None):
        p = self.params
        return ops.step_enc_spec(x, fp, p['fcs_w'], p['fcs_b'], p['fcp_w'], p['fcp_b'], self.nbrs, out=out, env=env, bits=bits)


class NCMultiAgentPolicy(BatchedPolicy):
    """NeurComm: s = [relu(x~ W_ob), relu(p~ W_fp), relu(m~ W_msg)] -> LSTM(3H) (agents/utils.py:182-208).
    m~ = neighbours' previous h, NOT done-masked (Q3: agents/utils.py:182-183)."""
    name = 'nc'
    k_wh, k_b, k_wx = 'wh_hid', 'hid_b', 'wx_hid'
    k_ob = 'w_ob'
    scope = 'nc/lstm_comm_%d'
    coupled = True                # messages: neighbours' h_{t-1} enter every step
    msg_kind = ops.MSG_GATHER_RELU
    enc_writes_bits = True

    def _phases(self):
        H, F, A = self.n_h, self.n_feat, self.n_a
        s = self.scope
        msg = [('w_msg', s + '/w_msg', (H * self.m_max, H), self._L_nbr(H)), ('w_msg_b', s + '/b_msg', (H,), self._L_nbr())]
        ob = [('w_ob', s + '/w_ob', (self.n_obs, H), self._L_slots), ('w_ob_b', s + '/b_ob', (H,), None)]
        fp = [('w_fp', s + '/w_fp', (self.n_na, H), self._L_fp), ('w_fp_b', s + '/b_fp', (H,), self._L_nbr())]
        hid = [('wx_hid', s + '/wx_hid', (3 * H, 4 * H), self._L_blocks(H, 3)), ('wh_hid', s + '/wh_hid', (H, 4 * H), None),
               ('hid_b', s + '/b_hid', (4 * H,), None)]
        # creation order: lstm_comm (agents/utils.py:141-162) msg, ob, fp; lstm_comm_hetero (258-281) ob, fp, msg
        first = ob + fp + msg if self.hetero else msg + ob + fp
        return [first + hid, self._head_phase(self.name + '/pi_%d', self.name + '/v_%d')]

    def _enc(self, xv, fp):
        """Observation + fingerprint thirds of s, already multiplied by their rows of wx_hid."""
        p = self.params
        H = self.n_h
        s = ops.fc_concat([self._ob_part(xv, 'w_ob', 'w_ob_b'), (fp, p['w_fp'], p['w_fp_b'], self.nbr_idx)], ops.BIAS_RELU)
        return ops.linear(s, p['wx_hid'][:, :2 * H])

    def _enc_saved(self, xv, fp, S, bits=None):
        """[hx | hp] as the rollout wrote it into the first 2H columns of the saved LSTM inputs (backward only); bits: its sign image."""
        p = self.params
        H = self.n_h
        Sv = S.view(self.N, -1, S.shape[-1])[:, :, :2 * H]
        return ops.fc_concat([self._ob_part(xv, 'w_ob', 'w_ob_b'), (fp, p['w_fp'], p['w_fp_b'], self.nbr_idx)], ops.BIAS_RELU, saved=Sv,
                             bits=bits)

    def _recur_in(self, enc, h):
        p = self.params
        H = self.n_h
        m = ops.nbr_gather(h, self.nbr_idx)                                   # un-masked previous h
        hm = torch.relu(torch.baddbmm(p['w_msg_b'].unsqueeze(1), m, p['w_msg']))
        return torch.baddbmm(enc, hm, p['wx_hid'][:, 2 * H:])

    def _enc_infer(self, xv, fp, out=None):
        """x-side mode: the full LSTM input [hx | hp | hm] [N,E,3H] (a fresh buffer or the given slot of the saved
        activations) with [hx | hp] filled in; the recurrent step adds the message third.  Else: [hx | hp] @ Wx[:2H]."""
        p = self.params
        H = self.n_h
        full = None
        if self.xside:
            full = out if out is not None else torch.empty(self.N, xv.shape[1], 3 * H, dtype=F32, device=xv.device)
        s = None if full is None else full[:, :, :2 * H]
        if self._enc_one_launch(xv, H):
            # [hx | hp] of agents/utils.py:199 by ONE kernel (fingerprint gather folded in)
            s = ops.fc_fwd_multi([self._ob_part(xv, 'w_ob', 'w_ob_b'), (fp, p['w_fp'], p['w_fp_b'], self.nbr_idx)],
                                 ops.BIAS_RELU, out=s)
        else:
            if s is None:
                s = torch.empty(self.N, xv.shape[1], 2 * H, dtype=F32, device=xv.device)
            self._fc_ob_infer(xv, 'w_ob', 'w_ob_b', ops.BIAS_RELU, out=s[:, :, :H])
            self._fc_infer(ops.nbr_gather(fp, self.nbr_idx), 'w_fp', 'w_fp_b', ops.BIAS_RELU, out=s[:, :, H:])
        return full if self.xside else torch.bmm(s, p['wx_hid'][:, :2 * H])

    def fused_env_encode(self, fp_next, out):
        return self._fused_spec(('w_ob', 'w_ob_b'), ('w_fp', 'w_fp_b'), fp_next, out) if self.xside else None

    def enc_in_kernel(self, E, compact):
        """The one-launch lock-step also runs the two input encoders (and, for the batched CACC engine, the env step): ONE launch per
        lock-step (csrc/lstm_mfma.hip <4,1,1>).  NMARL_NC_ONE_LAUNCH=0: the encoders stay behind the env kernel (nmarl_cacc_step_encode)."""
        return bool(compact) and self.xside and not self.hetero and self.pv_one_launch(E) and \
            ops.step_enc_supported(self.n_feat, self.n_a, self.m_max, self.n_fc, self.n_h, self.N) and \
            os.environ.get('NMARL_NC_ONE_LAUNCH', '1') != '0'

    @property
    def env_step_in_kernel(self):
        # lstm_comm: the one-launch form exists and is bit-identical (tests), but on the same box it is no faster than the env
        # kernel behind the <4,1,1> launch -- rollout graph 5.59-5.62 vs 5.53-5.56 ms (two launches, round 5: 5.60-5.64),
        # profiles/r06_ab_lockstep_nc.txt -- so the env kernel stays a launch of its own.  NMARL_NC_ENV_IN_KERNEL=1: one launch.
        return os.environ.get('NMARL_NC_ENV_IN_KERNEL', '0') == '1'

    def _enc_spec(self, x, fp, out, env=None, bits=None):
        p = self.params
        return ops.step_enc_spec(x, fp, p['w_ob'], p['w_ob_b'], p['w_fp'], p['w_fp_b'], self.nbrs, out=out, env=env, bits=bits)

    def _recur_addends(self, enc, h, second=False, save=None, fuse_msg=False):
        p = self.params
        H = self.n_h
        if fuse_msg and self._msg() is not None:
            # hm = relu(m~ W_msg + b) is computed by the step kernel itself; the policy step keeps it in the last third
            # of the LSTM input (the saved message term of the update), the value re-step discards it
            return None, None, (enc[:, :, :2 * H], p['wx_hid'], self._img, None, self._msg(out=None if second else enc[:, :, 2 * H:]))
        m = ops.nbr_gather(h, self.nbr_idx)                                   # un-masked previous h (Q3)
        if self.xside:
            # hm = relu(m~ W_msg + b) becomes the last third of the LSTM input: in place for the policy step (it is the
            # saved message term of the update), into a scratch third for the value re-step
            if second:
                hm2 = self._scratch('_hm2', h)
                self._fc_infer(m, 'w_msg', 'w_msg_b', ops.BIAS_RELU, out=hm2)
                return None, None, (enc[:, :, :2 * H], p['wx_hid'], self._img, hm2)
            self._fc_infer(m, 'w_msg', 'w_msg_b', ops.BIAS_RELU, out=enc[:, :, 2 * H:])
            return None, None, (enc, p['wx_hid'], self._img)
        hm = self._fc_infer(m, 'w_msg', 'w_msg_b', ops.BIAS_RELU)
        return torch.bmm(hm, p['wx_hid'][:, 2 * H:]), enc, None

    def _seq_args(self):
        p = self.params
        return 'nc', p['wx_hid'][:, 2 * self.n_h:], p['w_msg'], p['w_msg_b'], None, None


class IC3MultiAgentPolicy(BatchedPolicy):
    """CommNet ("IC3"): s = tanh(x~ W_ob + b) + mean_nbr(h_prev) W_msg + b_msg -> LSTM(H)
    (agents/utils.py:385-408)."""
    name = 'ic3'
    k_wh, k_b, k_wx = 'wh_hid', 'hid_b', 'wx_hid'
    k_ob = 'w_ob'
    scope = 'ic3/lstm_ic3_%d'
    coupled = True
    msg_kind = ops.MSG_MEAN_ADD

    def _phases(self):
        H, F = self.n_h, self.n_feat
        s = self.scope
        return [[('w_msg', s + '/w_msg', (H, H), self._L_nbr()),
                 ('w_msg_b', s + '/b_msg', (H,), self._L_nbr()),
                 ('w_ob', s + '/w_ob', (self.n_obs, H), self._L_slots),
                 ('w_ob_b', s + '/b_ob', (H,), None),
                 ('wx_hid', s + '/wx_hid', (H, 4 * H), None),
                 ('wh_hid', s + '/wh_hid', (H, 4 * H), None),
                 ('hid_b', s + '/b_hid', (4 * H,), None)],
                self._head_phase(self.name + '/pi_%d', self.name + '/v_%d')]

    def _enc(self, xv, fp):
        p = self.params
        return ops.fc_concat([self._ob_part(xv, 'w_ob', 'w_ob_b')], ops.BIAS_TANH)

    def _recur_in(self, enc, h):
        p = self.params
        mm = ops.nbr_mean(h, self.nbr_idx)                                    # un-masked previous h
        s = enc + torch.baddbmm(p['w_msg_b'].unsqueeze(1), mm, p['w_msg'])
        return torch.bmm(s, p['wx_hid'])

    def save_spec(self):
        # ENC: tanh(x~ W_ob + b) of every lock-step (the update's encoder backward needs no forward pass); MM: mean_nbr(h_{t-1}), the
        # message layer's input as the step kernel's pre-phase forms it (its weight gradient = MM^T D1: no averaging pass over the
        # h sequence in the update); one zero slab more, like the saved LSTM inputs
        return {'ENC': self.n_h, 'MM': self.n_h}

    save_pad = ('MM',)
    _mm_was_saved = False

    def _enc_saved(self, xv, fp, S):
        enc = getattr(self, '_extra', {}).get('ENC')
        if enc is None or not self._enc_was_saved:
            return self._enc(xv, fp)         # recomputed (one streaming pass)
        return ops.fc_concat([self._ob_part(xv, 'w_ob', 'w_ob_b')], ops.BIAS_TANH, saved=enc.view(self.N, -1, self.n_h))

    _enc_was_saved = False

    def _enc_infer(self, xv, fp, out=None):
        return self._fc_ob_infer(xv, 'w_ob', 'w_ob_b', ops.BIAS_TANH, out=out)

    def _x_target(self, h, second, save):
        """Where the LSTM input of this step goes: the slot of the saved activations (policy step of the batched
        engine), else a scratch tensor (one per half of the lock-step)."""
        if save is not None and not second:
            return save['S']
        return self._scratch('_x2' if second else '_x1', h)

    def _recur_addends(self, enc, h, second=False, save=None, fuse_msg=False):
        p = self.params
        if fuse_msg and self._msg() is not None:
            # s = mean_nbr(h) W_msg + b_msg + enc inside the step kernel; the policy step keeps it for the update
            keep = save is not None and not second
            if keep and 'MM' in save:
                self._mm_was_saved = True
                return None, None, (None, p['wx_hid'], self._img, None, self._msg(enc=enc, out=save['S'], mean_out=save['MM']))
            return None, None, (None, p['wx_hid'], self._img, None, self._msg(enc=enc, out=save['S'] if keep else None))
        if not second:
            self._mm_was_saved = False       # this path keeps no mean_nbr(h): the update averages the h sequence itself
        if self.xside:
            x = self._x_target(h, second, save)
            self._fc_infer(ops.nbr_mean(h, self.nbr_idx), 'w_msg', 'w_msg_b', ops.BIAS_NONE, out=x).add_(enc)
            return None, None, (x, p['wx_hid'], self._img)
        s = self._fc_infer(ops.nbr_mean(h, self.nbr_idx), 'w_msg', 'w_msg_b', ops.BIAS_NONE).add_(enc)
        return torch.bmm(s, p['wx_hid']), None, None

    def _seq_args(self):
        p = self.params
        return 'ic3', p['wx_hid'], p['w_msg'], p['w_msg_b'], None, None

    # -- the observation encoder inside the one-launch step (csrc/lstm_mfma.hip, OBENC)
    _ob_img = None
    _ob_pad = None

    def encodes_in_step(self, E, compact):
        return compact and not self.hetero and self.n_obs != self.n_feat and \
            ops.ob_encoder_supported(self.n_feat, self.n_obs, self.n_h) and self.pv_one_launch(E)

    def refresh_wimage(self):
        super().refresh_wimage()
        if self.xside and not self.hetero and ops.ob_encoder_supported(self.n_feat, self.n_obs, self.n_h):
            if self._ob_pad is None:
                self._ob_pad = torch.zeros(self.N, ops.FC_J, self.n_h, dtype=F32, device=self.device)
            self._ob_img = ops.lstm_ob_wimage(self.params['w_ob'], self._ob_pad, out=self._ob_img)

    def _ob_spec(self, ob):
        p = self.params
        return dict(x=ob, nbr=self.nbr_self, img=self._ob_img, b=p['w_ob_b'], w=p['w_ob'])


class ConsensusPolicy(LstmPolicy):
    """ConseNet ("IA2C_CU", policies.py:339-426): per agent fc(own obs) -> LSTM -> heads inside ONE
    optimiser; after every update each agent's LSTM weights are replaced by the mean over itself and its
    neighbours (`_get_critic_wts` averages only the `lstm_%da` scope)."""
    name = 'cu'
    k_ob = 'fc_w'

    def _phases(self):
        H, F = self.n_h, self.n_feat
        return [[('fc_w', 'cu/fc_%da/w', (F, H), None), ('fc_b', 'cu/fc_%da/b', (H,), None),
                 ('lstm_wx', 'cu/lstm_%da/wx', (H, 4 * H), None),
                 ('lstm_wh', 'cu/lstm_%da/wh', (H, 4 * H), None),
                 ('lstm_b', 'cu/lstm_%da/b', (4 * H,), None)] + self._head_phase('cu/pi_%da' if self.hetero else 'cu/pi_%d', 'cu/v_%da')]   # policies.py:381-390: the identical branch names the actor head pi_<i>, the hetero one pi_<i>a

    def _own(self, xv):
        return xv[:, :, :self.n_feat]        # the consensus net sees the agent's own features only (compact obs: all of xv)

    def _enc(self, xv, fp, saved=None):
        p = self.params
        s = ops.fc_concat([(self._own(xv), p['fc_w'], p['fc_b'])], ops.BIAS_RELU, saved=saved)
        return s if self.xside else ops.linear(s, p['lstm_wx'])

    def _enc_infer(self, xv, fp, out=None):
        s = self._fc_infer(self._own(xv), 'fc_w', 'fc_b', ops.BIAS_RELU, out=out)
        return s if self.xside else torch.bmm(s, self.params['lstm_wx'])

    def consensus_update(self):
        """policies.py:357-364, 403-426: simultaneous neighbourhood average of (wx, wh, b) of every agent's LSTM.
        The three tensors are adjacent in the flat buffer: ONE [N,N] x [N,K] product."""
        ps = self.params
        o0 = ps.index['lstm_wx'][0]
        o1 = ps.index['lstm_b'][0] + ps.index['lstm_b'][1]
        if not hasattr(self, '_avg'):
            tab = self.nbr_idx.cpu().numpy()
            A = np.eye(self.N, dtype=np.float32)
            for i in range(self.N):
                for j in tab[i]:
                    if j >= 0:
                        A[i, j] = 1.0
            self._avg = torch.from_numpy(A / A.sum(1, keepdims=True)).to(self.device)
        with torch.no_grad():
            blk = ps.flat[:, o0:o1]
            blk.copy_(self._avg @ blk)


class DIALMultiAgentPolicy(BatchedPolicy):
    """DIAL (policies.py:479-525 + lstm_dial agents/utils.py:515-599):
    s_i = relu(x~_i W_ob) + relu([mfc_j(h_j) for j in nbr(i)] W_msg) + onehot_H(argmax pi_i(t-1)) -> LSTM(H);
    the message encoder mfc_j = relu(h_j W + b) acts on the sender's un-masked previous h."""
    bptt_takes_head_dy = False      # (its reverse recurrence is step-wise launches taking dL/dh as a tensor: the fused pass writes it)

    name = 'dial'
    k_wh, k_b, k_wx = 'wh_hid', 'hid_b', 'wx_hid'
    k_ob = 'w_ob'
    coupled = True
    msg_kind = ops.MSG_DIAL
    msg_inplace_ok = True

    def _phases(self):
        H, F = self.n_h, self.n_feat
        s = 'dial/lstm_comm_%d'
        return [[('w_msg', s + '/w_msg', (H * self.m_max, H), self._L_nbr(H)),
                 ('w_msg_b', s + '/b_msg', (H,), self._L_nbr()),
                 ('w_ob', s + '/w_ob', (self.n_obs, H), self._L_slots),
                 ('w_ob_b', s + '/b_ob', (H,), None),
                 ('wx_hid', s + '/wx_hid', (H, 4 * H), None),
                 ('wh_hid', s + '/wh_hid', (H, 4 * H), None),
                 ('hid_b', s + '/b_hid', (4 * H,), None)],
                [('mfc_w', 'dial/mfc_%d/w', (H, H), None), ('mfc_b', 'dial/mfc_%d/b', (H,), None)],
                self._head_phase('dial/pi_%d', 'dial/v_%d')]

    def _ai_scale(self, fp):
        """lstm_dial_hetero (agents/utils.py:676-688): an agent without neighbours gets neither messages nor `ai`."""
        if not self.hetero:
            return None
        if not hasattr(self, '_has_nbr'):
            self._has_nbr = torch.tensor([float(self._m(i) > 0) for i in range(self.N)], dtype=fp.dtype,
                                         device=fp.device).view(self.N, 1, 1)
        return self._has_nbr

    def _own_action_onehot(self, fp):
        """one_hot(argmax(p_i), n_h) (agents/utils.py:577): first maximum, like tf.argmax."""
        oh = torch.nn.functional.one_hot(torch.argmax(fp, dim=-1), self.n_h).to(fp.dtype)
        sc = self._ai_scale(fp)
        return oh if sc is None else oh * sc

    def _enc(self, xv, fp):
        p = self.params
        return ops.fc_concat([self._ob_part(xv, 'w_ob', 'w_ob_b')], ops.BIAS_RELU) + self._own_action_onehot(fp)

    def _recur_in(self, enc, h):
        p = self.params
        msg = torch.relu(torch.baddbmm(p['mfc_b'].unsqueeze(1), h, p['mfc_w']))        # sender side, un-masked h
        hm = torch.relu(torch.baddbmm(p['w_msg_b'].unsqueeze(1), ops.nbr_gather(msg, self.nbr_idx), p['w_msg']))
        return torch.bmm(enc + hm, p['wx_hid'])

    def _enc_saved(self, xv, fp, S):
        # recomputed (only the recurrence is saved) for the observation layer's backward; CoupledSequenceSaved never reads the
        # VALUE of enc (the LSTM inputs S = enc + hm are saved), and the own-action one-hot is a constant: left out here
        return ops.fc_concat([self._ob_part(xv, 'w_ob', 'w_ob_b')], ops.BIAS_RELU)

    def _enc_infer(self, xv, fp, out=None):
        sc = self._ai_scale(fp)
        return ops.onehot_argmax_add_(self._fc_ob_infer(xv, 'w_ob', 'w_ob_b', ops.BIAS_RELU), fp, None if sc is None else sc.view(-1))

    def save_spec(self):
        return {'A1': self.n_h, 'A2': self.n_h}        # hm (post-relu), msg (post-relu): relu masks of the backward

    _x_target = IC3MultiAgentPolicy._x_target

    # -- the sender layer msg = relu(h W_mfc + b) of the NEW h in the policy step's epilogue (csrc/lstm_mfma.hip NXT): the value
    # re-step and the next lock-step's policy step gather it as it is -- no fc launch on h in between
    save_next = ('A2',)       # save slots the policy step of lock-step t fills for lock-step t + 1 (models.enable_saved_acts)
    _mfc_img = None
    _m_next = None            # (pointer, version, message vectors) of the h the last policy step produced
    _h_out = None
    _nxt = None

    def refresh_wimage(self):
        super().refresh_wimage()
        if self._msg_img is not None:
            self._mfc_img = ops.lstm_msg_wimage(self.params['mfc_w'], out=self._mfc_img)
        self._m_next = None   # the weights may have changed: vectors of the old sender layer are stale

    def invalidate_cached_msg(self):
        """Entry points that write the recurrent state OUT OF BAND (kernels through raw pointers do not bump a tensor's version
        counter: models.reset_states, the batch hand-over of BatchedTrainer) call this: the next policy step recomputes the senders'
        message vectors instead of trusting the pointer / version match below."""
        self._m_next = None

    def _cached_msg(self, h):
        """The message vectors of exactly this h, if the last policy step left them: same memory, and no in-place torch op on it
        since (version counter; kernels writing through raw pointers are the policy steps themselves -- every other writer calls
        `invalidate_cached_msg`)."""
        m = self._m_next
        if m is not None and m[0] == h.data_ptr() and m[1] == h._version and m[2].shape == h.shape:
            return m[2]
        return None

    def step_policy(self, enc, h, c, done, h_out, c_out, pi_out, act_out, done_is_zero=False, gates=None, save=None, **draw):
        self._nxt, self._h_out = None, h_out
        try:
            r = super().step_policy(enc, h, c, done, h_out, c_out, pi_out, act_out, done_is_zero, gates=gates, save=save, **draw)
        finally:
            self._h_out = None
        self._m_next = None if self._nxt is None else (h_out.data_ptr(), h_out._version, self._nxt)
        return r

    def _recur_addends(self, enc, h, second=False, save=None, fuse_msg=False):
        p = self.params
        keep = save is not None and not second
        fused = fuse_msg and self._msg() is not None
        msg = self._cached_msg(h) if fused else None
        if msg is None:
            msg = self._fc_infer(h, 'mfc_w', 'mfc_b', ops.BIAS_RELU, out=save['A2'] if keep else None)
        elif keep and msg.data_ptr() != save['A2'].data_ptr():
            save['A2'].copy_(msg)
        if fused:
            # hm = relu([msg_j] W_msg + b) and s = hm + enc inside the step kernel (csrc/lstm_mfma.hip MSG 3): no gather / GEMM /
            # bias-activation / add launches; the policy step keeps hm (relu mask of the backward) and s (the LSTM input)
            m = self._msg(src=msg, enc=enc, out=save['S'] if keep else None, out2=save['A1'] if keep else None)
            if not second and self._h_out is not None and self._mfc_img is not None:
                nxt = save.get('A2_next') if keep else None
                if nxt is None:
                    nxt = self._scratch('_mn0', h)
                    if nxt.data_ptr() == msg.data_ptr():
                        nxt = self._scratch('_mn1', h)
                m['next'] = dict(img=self._mfc_img, b=p['mfc_b'], w=p['mfc_w'], out=nxt)
                self._nxt = nxt
            return None, None, (None, p['wx_hid'], self._img, None, m)
        hm = self._fc_infer(ops.nbr_gather(msg, self.nbr_idx), 'w_msg', 'w_msg_b', ops.BIAS_RELU,
                            out=save['A1'] if keep else None)
        if self.xside:
            x = self._x_target(h, second, save)
            torch.add(hm, enc, out=x)
            return None, None, (x, p['wx_hid'], self._img)
        return torch.bmm(hm.add_(enc), p['wx_hid']), None, None

    def _seq_args(self):
        p = self.params
        return 'dial', p['wx_hid'], p['w_msg'], p['w_msg_b'], p['mfc_w'], p['mfc_b']
