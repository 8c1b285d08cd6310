"""IA2C / MA2C algorithm classes on MI355X -- the reference's `agents.models` surface
(models.py:15-309) over a batched, device-resident engine.

Each class keeps the reference's constructor and methods
    forward / add_transition / backward / reset / save / load
with the reference's E = 1, NumPy-in / NumPy-out semantics (so the reference's
Trainer and tests drive it unchanged), and adds the batched API the MI355X
Trainer uses for E >> 1 replicas with everything resident in HBM:
    act()  record()  bootstrap()  update()  reset_states()

Engine (per n_step batch, E replicas, N agents):
  rollout   2 recurrent steps per lock-step, reproducing the reference's quirk Q1
            (utils.py:129-151 + policies.py:119-134: `forward('v')` re-steps the LSTM
            from the state `forward('p')` just wrote);
  buffers   [T,...] device tensors replacing OnPolicyBuffer's Python lists
            (agents/utils.py:722-761, 819-835);
  update    nmarl_nstep_return -> autograd unroll -> A2C loss (policies.py:20-30,
            232-255; means over T*E, sum over agents) -> [RCCL all-reduce] ->
            nmarl_rmsprop_tf_clip (policies.py:32-39).
"""
import logging
import os

import numpy as np
import torch

from .. import ops
from .policies import (ConsensusPolicy, DIALMultiAgentPolicy, FPPolicy, IC3MultiAgentPolicy, LstmPolicy,
                       NCMultiAgentPolicy)
from .utils import Scheduler

F32 = torch.float32


class IA2C:
    """Independent A2C with per-agent optimisers (models.py:15-158)."""

    policy_cls = LstmPolicy
    per_agent_optimizer = True       # IA2C: one RMSProp + one clip norm per agent (models.py:148-152)
    uses_fingerprint = False         # forward() receives neighbour policies `ps` (MA2C family)

    def __init__(self, n_s_ls, n_a_ls, neighbor_mask, distance_mask, coop_gamma, total_step, model_config,
                 seed=0, num_envs=1, device='cuda', dist_group=None, n_feat=None, n_feat_ls=None, obs_order=None):
        """The reference's constructor (models.py:20-24) + batching arguments.  obs_order (IA2C family): the env's
        `neighbor_order`, see BatchedPolicy.  n_feat_ls: per-agent OWN observation
        widths of heterogeneous systems (agents with different action counts, `identical_agent = False`,
        models.py:89-96).  MA2C envs report them as n_s_ls; the IA2C family's n_s_ls are the concatenated widths, from
        which the own widths are recovered when the system (I + neighbor_mask) x = n_s_ls determines them."""
        self.name = getattr(self, 'name', 'ia2c')
        self._init_algo(n_s_ls, n_a_ls, neighbor_mask, distance_mask, coop_gamma, total_step, seed,
                        model_config, num_envs, device, dist_group, n_feat, n_feat_ls, obs_order)

    # ------------------------------------------------------------------ construction
    def _init_algo(self, n_s_ls, n_a_ls, neighbor_mask, distance_mask, coop_gamma, total_step, seed,
                   model_config, num_envs, device, dist_group, n_feat, n_feat_ls=None, obs_order=None):
        self.n_s_ls, self.n_a_ls = [int(x) for x in n_s_ls], [int(x) for x in n_a_ls]
        self.n_a = max(self.n_a_ls)
        self.neighbor_mask = np.asarray(neighbor_mask)
        self.distance_mask = np.asarray(distance_mask)
        self.n_agent = len(self.neighbor_mask)
        self.identical_agent = max(self.n_a_ls) == min(self.n_a_ls)          # models.py:89-96
        self.n_feat_ls = None
        if not self.identical_agent:
            self.n_feat_ls = self._own_widths(n_feat_ls)
            n_feat = max(self.n_feat_ls)
        self.coop_gamma = float(coop_gamma)
        self.reward_clip = model_config.getfloat('reward_clip')
        self.reward_norm = model_config.getfloat('reward_norm')
        self.n_step = model_config.getint('batch_size')
        self.n_fc = model_config.getint('num_fc')
        self.n_lstm = model_config.getint('num_lstm')
        self.E = int(num_envs)
        self.device = torch.device(device)
        self.dist_group = dist_group
        self.world_size = 1
        if dist_group is not None:
            import torch.distributed as dist
            self.world_size = dist.get_world_size(dist_group)
        self.seed = seed
        m = self.neighbor_mask.sum(axis=1)
        if n_feat is None:
            # MA2C envs report the own-feature width; IA2C envs report n_feat*(1+m_i)
            n_feat = self.n_s_ls[0] if self._is_ma2c() else self.n_s_ls[0] // (1 + int(m[0]))
        self.n_feat = int(n_feat)
        self.policy = self.policy_cls(self.n_feat, self.n_a, self.neighbor_mask, n_fc=self.n_fc,
                                      n_h=self.n_lstm, device=self.device, n_feat_ls=self.n_feat_ls,
                                      n_a_ls=None if self.identical_agent else self.n_a_ls,
                                      obs_order=None if self._is_ma2c() else obs_order)
        self.policy.params.init_reference_order()       # consumes np.random like the reference's ortho_init
        self.n_s = self.n_s_ls[0]
        N, E, H, T = self.n_agent, self.E, self.n_lstm, self.n_step
        d = self.device
        z = lambda *s: torch.zeros(*s, dtype=F32, device=d)      # noqa: E731
        # recurrent state [c,h] of policies.py:151-154; static buffers (hipGraph-friendly), updated in place
        self.h_fw, self.c_fw, self.h_bw, self.c_bw = z(N, E, H), z(N, E, H), z(N, E, H), z(N, E, H)
        self._h2, self._c2 = z(N, E, H), z(N, E, H)        # scratch of the value re-step (quirk Q1)
        # a fresh episode's fingerprint: uniform over the agent's OWN actions (cacc_env.py:184, atsc_env.py:498-499)
        fp0 = np.zeros((N, 1, self.n_a), dtype=np.float32)
        for i in range(N):
            fp0[i, 0, :self.n_a_ls[i]] = 1.0 / self.n_a_ls[i]
        self.fp_uniform = torch.from_numpy(fp0).to(d)
        self._fp_eval = self.fp_uniform.expand(N, E, self.n_a).clone()
        self.t = 0
        self.total_step = total_step
        self.sess = None                                  # the reference Trainer reads model.sess (TF leak)
        if total_step:
            self._init_train(model_config, self.distance_mask, coop_gamma)

    def _is_ma2c(self):
        return self.name.startswith('ma2c')

    def _own_widths(self, n_feat_ls):
        """Own observation width of every agent of a heterogeneous system."""
        if n_feat_ls is not None:
            return [int(x) for x in n_feat_ls]
        if self._is_ma2c():
            return list(self.n_s_ls)                       # MA2C envs hand over the own features only
        a = np.eye(self.n_agent) + (self.neighbor_mask == 1)
        if abs(np.linalg.det(a)) < 1e-9:
            raise ValueError('heterogeneous IA2C: own observation widths are not determined by n_s_ls; pass n_feat_ls')
        x = np.linalg.solve(a, np.asarray(self.n_s_ls, dtype=np.float64))
        if np.abs(x - np.round(x)).max() > 1e-6 or x.min() < 1:
            raise ValueError('heterogeneous IA2C: n_s_ls is not a sum of own + neighbour widths; pass n_feat_ls')
        return [int(round(v)) for v in x]

    def _pad_policies(self, ps):
        """list of N probability vectors (ragged for heterogeneous agents, models.py:229-236) -> [N,1,A] f32."""
        out = np.zeros((self.n_agent, 1, self.n_a), dtype=np.float32)
        for i in range(self.n_agent):
            v = np.asarray(ps[i], dtype=np.float32).reshape(-1)
            out[i, 0, :len(v)] = v
        return torch.from_numpy(out).to(self.device)

    def _unpad_policies(self, pi):
        """[N,A] -> what the reference returns: an [N,A] array, or a list of ragged vectors (policies.py:314-316)."""
        if self.identical_agent:
            return pi
        return [pi[i, :self.n_a_ls[i]] for i in range(self.n_agent)]

    @property
    def fp(self):
        """Current fingerprints (previous-step policies) [N,E,A]: slot t of the rollout buffer while
        training, a standalone tensor for evaluation-only models (total_step = 0)."""
        return self.buf_fp[self.t] if self.total_step else self._fp_eval

    def _init_scheduler(self, model_config):
        lr_init = model_config.getfloat('lr_init')
        lr_decay = model_config.get('lr_decay')
        if lr_decay == 'constant':
            self.lr_scheduler = Scheduler(lr_init, decay=lr_decay)
        else:
            self.lr_scheduler = Scheduler(lr_init, model_config.getfloat('lr_min'), self.total_step, decay=lr_decay)

    def _init_train(self, model_config, distance_mask, coop_gamma):
        self._init_scheduler(model_config)
        self.v_coef = model_config.getfloat('value_coef')
        self.e_coef = model_config.getfloat('entropy_coef')
        self.max_grad_norm = model_config.getfloat('max_grad_norm')
        self.rmsp_alpha = model_config.getfloat('rmsp_alpha')
        self.rmsp_epsilon = model_config.getfloat('rmsp_epsilon')
        self.gamma = model_config.getfloat('gamma')
        N, E, T, d = self.n_agent, self.E, self.n_step, self.device
        p = self.policy
        # rollout buffers (replace OnPolicyBuffer's lists); every slot [t] is contiguous so kernels and
        # GEMMs write their results straight into it.  buf_x / buf_fp carry T+1 slots: slot t+1 receives
        # the observation / policy produced at lock-step t, slot T is the bootstrap input and becomes
        # slot 0 of the next batch.
        self.buf_x = torch.zeros(T + 1, E, N, p.n_obs, dtype=F32, device=d)
        self.buf_fp = self.fp_uniform.expand(T + 1, N, E, self.n_a).clone()
        self._pi_boot, self._v_boot = torch.zeros(N, E, self.n_a, dtype=F32, device=d), torch.zeros(N, E, dtype=F32, device=d)
        self.buf_act = torch.zeros(T, E, N, dtype=torch.uint8, device=d)
        self.buf_v = torch.zeros(T, N, E, dtype=F32, device=d)
        self.buf_done_pre = torch.zeros(T, E, dtype=F32, device=d)
        self.buf_done_post = torch.zeros(T, E, dtype=torch.uint8, device=d)
        spatial = self.coop_gamma >= 0
        self.buf_r = torch.zeros((T, E, N) if spatial else (T, E), dtype=F32, device=d)
        self.R = torch.zeros(N, T, E, dtype=F32, device=d)
        self.Adv = torch.zeros(N, T, E, dtype=F32, device=d)
        self.dist_dev = torch.from_numpy(self.distance_mask.astype(np.int32)).to(d)
        self.t = 0
        self.grad_norm = torch.zeros(N, dtype=F32, device=d)
        self.last_loss = None
        # steps whose pre-step done flag may be non-zero (None = any, the reference API); the batched trainer
        # sets (0,) because episodes start only at batch boundaries (quirk Q4)
        self.masked_steps = None
        self.save_acts = False

    def enable_saved_activations(self):
        """Batched engine, uncoupled nets: the rollout's policy steps ARE the forward pass of the update (on-policy
        A2C: same weights, same inputs, states_bw = the state the rollout started from), so the step kernel saves the
        LSTM inputs, gates and state sequences and update() runs the backward only.  The critic's neighbour-action
        term is added for all T lock-steps by one launch at update time.  Returns False if the policy cannot do it."""
        p = self.policy
        if not p.can_save_acts:
            return False
        N, E, T, H, d = self.n_agent, self.E, self.n_step, self.n_lstm, self.device
        KX = p.params[p.k_wx].shape[1]
        # one zero slab more than needed, so that [S | .] and the state sequences have the SAME (T + 1)-slab shape -- the
        # update's weight-gradient GEMMs then read the saved buffers in place (ops._lstm_seq_x_backward, agents/sequence.py)
        self.S_ext = torch.zeros(N, T + 1, E, KX, dtype=F32, device=d)
        self.S_buf = self.S_ext[:, :T]
        self.G_buf = torch.zeros(N, T, E, 4 * H, dtype=F32, device=d)
        self.H_all = torch.zeros(N, T + 1, E, H, dtype=F32, device=d)
        self.C_all = torch.zeros(N, T + 1, E, H, dtype=F32, device=d)
        self.buf_vn = torch.zeros(N, T, E, dtype=F32, device=d)          # agent-major values (critic's h part)
        # coupled nets: further per-step message terms the backward needs (policy.save_spec)
        # (`save_next` keys: one slab more -- the policy step of lock-step t fills slot t + 1, e.g. lstm_dial's message vectors)
        # (`save_pad` keys: one ZERO slab more, like S_ext -- the update's weight-gradient GEMM reads the (T + 1)-slab buffer in place)
        nxt, pad = getattr(p, 'save_next', ()), getattr(p, 'save_pad', ())
        full = {k: torch.zeros(N, T + 1 if (k in nxt or k in pad) else T, E, w, dtype=F32, device=d) for k, w in p.save_spec().items()}
        p._extra = {k: v[:, :T] for k, v in full.items()}
        p._extra_full = full
        self._extra_next = {k: full[k] for k in nxt if k in full}
        self.save_acts = True
        return True

    def enable_compact_obs(self):
        """Batched engine: the env writes compact observations [E,N,n_feat] (own features only); the rollout's encoder
        gathers the neighbours inside its kernels, forward (rollout) and backward (update)."""
        p = self.policy
        if p.hetero or p.n_obs == p.n_feat:
            return False
        if p.params[p.k_ob].shape[2] != ops.FC_J or p.n_obs > ops.FC_MAX_F:
            return False          # the gathering encoder kernel (nmarl_fc_fwd_multi) is 64 outputs wide, inputs <= 64
        T, E, N = self.n_step, self.E, self.n_agent
        self.buf_x = torch.zeros(T + 1, E, N, p.n_feat, dtype=F32, device=self.device)
        self.compact_obs = True
        return True

    compact_obs = False

    def _save_slots(self, t):
        """Slots of lock-step t in the saved activations, for the policy step of a coupled net."""
        d = {k: v[:, t] for k, v in self.policy._extra.items()}
        d['S'] = self.S_buf[:, t]
        for k, v in getattr(self, '_extra_next', {}).items():
            d[k + '_next'] = v[:, t + 1]
        return d

    # ------------------------------------------------------------------ batched engine
    def reset_states(self, mask=None):
        """Policy._reset (policies.py:151-154, 334-336) for all replicas or those with mask != 0;
        also restores the uniform fingerprint of a fresh episode (cacc_env.py:184)."""
        self.policy.invalidate_cached_msg()
        if mask is None:
            for s in (self.h_fw, self.c_fw, self.h_bw, self.c_bw):
                s.zero_()
            self.fp.copy_(self.fp_uniform.expand_as(self.fp))
        else:
            keep = (mask == 0).to(F32).view(1, -1, 1)
            for s in (self.h_fw, self.c_fw, self.h_bw, self.c_bw):
                s.mul_(keep)
            self.fp.mul_(keep).add_((1.0 - keep) * self.fp_uniform)

    def _policy_step(self, obs, done, done_is_zero=False):
        """forward('p'): advances states_fw (policies.py:119-134); returns the pi LOGITS' softmax."""
        self.policy.refresh_wimage()          # reference API: the weights may have changed since the last call
        self._enc = self.policy.encode(obs, self.fp)
        self.policy.step(self._enc, self.h_fw, self.c_fw, done, self.h_fw, self.c_fw, done_is_zero)
        with torch.no_grad():
            return self.policy.pi(self.h_fw)

    def _value_step(self, obs, done, na_onehot, done_is_zero=False, out=None, reuse_enc=False):
        """forward('v'): re-steps the LSTM from the state forward('p') wrote (quirk Q1), without
        storing the result (policies.py:124-133).  `reuse_enc`: obs / fingerprints are those of the
        preceding _policy_step, whose encoding is shared."""
        if not reuse_enc:
            self.policy.refresh_wimage()
        enc = self._enc if reuse_enc else self.policy.encode(obs, self.fp)
        self.policy.step(enc, self.h_fw, self.c_fw, done, self._h2, self._c2, done_is_zero, second=True)
        with torch.no_grad():
            return self.policy.value(self._h2, na_onehot, out=out)

    def encode_target(self, t):
        """Where lock-step t's LSTM input encoding lives when the env kernel produces it behind its step (fused encode):
        slot t of the saved activations, or a scratch buffer for the bootstrap step t = n_step (slab n_step of the saved
        inputs must stay zero: it pads the update's weight-gradient GEMMs)."""
        if t < self.n_step:
            return self.S_buf[:, t]
        if getattr(self, '_enc_boot', None) is None:
            self._enc_boot = torch.zeros_like(self.S_buf[:, 0])
        return self._enc_boot

    def act(self, done, mode=ops.SAMPLE_PHILOX, u=None, seed=0, env_id_base=0, step=0, step_dev=None,
            done_is_zero=False, pre_encoded=False, env_step=None):
        """One lock-step decision for all replicas at buffer slot t: reads the observation buf_x[t]
        and fingerprints buf_fp[t]; writes the action into buf_act[t], the value into buf_v[t] and the
        new policy into buf_fp[t+1] (env.update_fingerprint, utils.py:173).  done [E] f32 is the pre-step
        flag.  Two kernels after the encoder when the heads fuse (H = 64).  Returns the action slot (input
        of the env kernel)."""
        t = self.t
        p = self.policy
        ob = None
        if t == 0:
            p.refresh_wimage()                             # weights change between batches only (inside the hipGraph: one
            if self.save_acts:                             # small node)
                self.H_all[:, 0].copy_(self.h_fw)
                self.C_all[:, 0].copy_(self.c_fw)
        # enc is shared by the policy step and the value re-step (Q1)
        if pre_encoded:             # the env kernel of the previous lock-step already encoded buf_x[t] / fp[t] into slot t
            enc = self.S_buf[:, t]
        elif self.save_acts and p.enc_in_kernel(self.E, self.compact_obs):
            # the lock-step kernel runs both input encoders itself, from the compact observation and the fingerprints of slot t,
            # and leaves the LSTM input in slot t of the saved activations: no encoder launch at all
            # (env_step: the same launch also steps the env with the actions it draws -- CACCBatchEnv.inkernel_step)
            enc, ob = self.S_buf[:, t], dict(x=self.buf_x[t], fp=self.fp, env=env_step, bits=self._relu_bits(t))
        elif self.save_acts and 'ENC' in p._extra:
            # nets whose encoder output is NOT the LSTM input itself (CommNet: s = enc + message term): kept per lock-step so
            # that the update's encoder backward needs no forward pass.  Where the one-launch step runs the encoder too, `enc`
            # is only the slot its output goes to
            if p.encodes_in_step(self.E, self.compact_obs):
                # (env_step: the launch also steps the grid env, on compute units its LSTM blocks leave idle -- LargeGridBatchEnv.inkernel_step)
                enc, ob = p._extra['ENC'][:, t], (self.buf_x[t] if env_step is None else dict(x=self.buf_x[t], genv=env_step))
            else:
                enc = p.encode(self.buf_x[t], self.fp, out=p._extra['ENC'][:, t])
            p._enc_was_saved = True
        else:
            enc = p.encode(self.buf_x[t], self.fp, out=self.S_buf[:, t]) if self.save_acts else p.encode(self.buf_x[t], self.fp)
        if env_step is not None and (ob is None or not isinstance(ob, dict)):
            raise ValueError('env_step needs the lock-step kernel that runs the input encoders itself (policy.enc_in_kernel / encodes_in_step)')
        draw = dict(mode=mode, u=u, seed=seed, env_id_base=env_id_base, step=step, step_dev=step_dev)
        if self.save_acts and p.pv_one_launch(self.E):
            # the policy step reads slot t of the state sequences and writes slot t + 1, gates into G[:, t]; the value
            # (critic's h part) goes to the agent-major buffer, its neighbour-action term is added in update().  Coupled nets:
            # the policy step's message terms go to their save slots, the re-step's come from the new states of ALL agents
            p.step_policy_value(enc, self.H_all[:, t], self.C_all[:, t], done, self.buf_fp[t + 1], self.buf_act[t],
                                self.buf_vn[:, t], h_out=self.H_all[:, t + 1], c_out=self.C_all[:, t + 1],
                                gates=self.G_buf[:, t], defer_action_term=True,
                                **(dict(save=self._save_slots(t), ob=ob, carry=self._msg_carry(t)) if p.coupled else
                                   (dict(ob=ob) if ob is not None else {})),
                                **draw)
            return self.buf_act[t]
        if self.save_acts:
            # coupled nets: policy step (saves its message terms, gates, states), then the value re-step from the new
            # states of ALL agents (its message term is recomputed from them; nothing of it is kept)
            p.step_policy(enc, self.H_all[:, t], self.C_all[:, t], done, self.H_all[:, t + 1], self.C_all[:, t + 1],
                          self.buf_fp[t + 1], self.buf_act[t], done_is_zero, gates=self.G_buf[:, t], save=self._save_slots(t),
                          **draw)
            p.step_value(enc, self.H_all[:, t + 1], self.C_all[:, t + 1], done, self._h2, self._c2, self.buf_act[t],
                         self.buf_v[t], done_is_zero)
            return self.buf_act[t]
        if p.fused_pv:
            p.step_policy_value(enc, self.h_fw, self.c_fw, done, self.buf_fp[t + 1], self.buf_act[t], self.buf_v[t], **draw)
            return self.buf_act[t]
        p.step_policy(enc, self.h_fw, self.c_fw, done, self.h_fw, self.c_fw, self.buf_fp[t + 1], self.buf_act[t],
                      done_is_zero, **draw)
        p.step_value(enc, self.h_fw, self.c_fw, done, self._h2, self._c2, self.buf_act[t], self.buf_v[t], done_is_zero)
        return self.buf_act[t]

    _carry_buf = None
    _carry_holds = -1          # the lock-step whose policy-step message term `_carry_buf` holds (written by the re-step of the one before)

    def _msg_carry(self, t):
        """Coupled nets, one-launch lock-step t (t = n_step: the bootstrap step): the re-step's message term -- computed from the
        neighbours' new, un-masked h (quirk Q3) -- IS lock-step t + 1's policy-step message term, so launch t leaves it in
        `_carry_buf` (and CommNet's mean rows in slot t + 1 of the saved means) and launch t + 1 starts from it: no neighbour rows,
        no product in front of its K loop.  Lock-step 0 computes its own (the batch boundary reset finished replicas' states)."""
        p = self.policy
        if not p.coupled or getattr(p, 'msg_kind', 0) not in (ops.MSG_GATHER_RELU, ops.MSG_MEAN_ADD) or \
                os.environ.get('NMARL_MSG_CARRY', '1') == '0':
            return None
        if self._carry_buf is None:
            self._carry_buf = torch.zeros(self.n_agent, self.E, self.n_lstm, dtype=F32, device=self.device)
        d = dict(carry_in=self._carry_buf if (t > 0 and self._carry_holds == t) else None,
                 carry_out=self._carry_buf if t < self.n_step else None)
        mm = getattr(p, '_extra_full', {}).get('MM')
        if mm is not None and t + 1 < self.n_step and p.msg_kind == ops.MSG_MEAN_ADD:
            d['mean_next'] = mm[:, t + 1]
        self._carry_holds = t + 1 if t < self.n_step else -1
        return d

    S_bits = None

    def _relu_bits(self, t):
        """Slot t of the sign image of the saved LSTM inputs ([N,T,E,4] int32, ops.relu_bits_pack's layout): written by the
        lock-step kernel that runs the input encoders, read by the update's encoder backward instead of S itself."""
        if os.environ.get('NMARL_FC_BWD_PAIR', '1') == '0' or not getattr(self.policy, 'enc_writes_bits', False):
            return None
        if self.S_bits is None:
            self.S_bits = torch.zeros(self.n_agent, self.n_step, self.E, 4, dtype=torch.int32, device=self.device)
        p = self.policy
        p._bits_steps = 1 if t == 0 else p._bits_steps + 1
        return self.S_bits[:, t]

    def record(self, reward, done_post):
        """model.add_transition's reward path (models.py:26-32) for slot t: normalise, clip, store."""
        t = self.t
        self.buf_r[t].copy_(self._norm_reward(reward))
        self.buf_done_post[t].copy_(done_post)
        self.t = t + 1

    def _norm_reward(self, r):
        if self.reward_norm > 0:
            r = r / self.reward_norm
        if self.reward_clip > 0:
            r = torch.clamp(r, -self.reward_clip, self.reward_clip)
        return r

    def load_rewards(self, raw_rewards):
        """Batched path: the env kernel wrote raw rewards for all T slots (and done flags straight
        into buf_done_post); normalise them in one pass and mark the batch complete."""
        # (written in place: `buf_r.copy_(temporary)` is a memcpy node inside the captured update)
        if self.reward_norm > 0:
            torch.div(raw_rewards, self.reward_norm, out=self.buf_r)
        else:
            torch.mul(raw_rewards, 1.0, out=self.buf_r)
        if self.reward_clip > 0:
            self.buf_r.clamp_(-self.reward_clip, self.reward_clip)
        self.t = self.n_step

    def bootstrap(self, done, action_scratch, mode=ops.SAMPLE_PHILOX, u=None, seed=0, env_id_base=0,
                  step=0, step_dev=None, done_is_zero=False, pre_encoded=False):
        """R for the unfinished replicas (utils.py:192-196) from buf_x[T] / buf_fp[T]: one more policy
        step (which advances states_fw -- quirk Q2) and the double-stepped value."""
        assert self.t == self.n_step
        p = self.policy
        ob = None
        if pre_encoded:
            enc = self.encode_target(self.n_step)
        elif self.save_acts and p.enc_in_kernel(self.E, self.compact_obs):
            # (nothing of the bootstrap step is kept; a coupled net's encoders hand their output to the K loop THROUGH the x slot:
            # a scratch one here, slab n_step of the saved inputs must stay zero)
            enc, ob = (self.encode_target(self.n_step) if p.coupled else None), dict(x=self.buf_x[self.n_step], fp=self.fp)
        elif self.save_acts and p.encodes_in_step(self.E, self.compact_obs):
            enc, ob = self.encode_target(self.n_step), self.buf_x[self.n_step]     # the step kernel runs the encoder itself
        else:
            enc = p.encode(self.buf_x[self.n_step], self.fp)
        if self.save_acts and p.pv_one_launch(self.E):     # from slot T of the sequences into the persistent state
            T = self.n_step
            p.step_policy_value(enc, self.H_all[:, T], self.C_all[:, T], done, self._pi_boot, action_scratch, self._v_boot,
                                h_out=self.h_fw, c_out=self.c_fw, mode=mode, u=u, seed=seed, env_id_base=env_id_base,
                                step=step, step_dev=step_dev, **(dict(ob=ob) if ob is not None else {}),
                                **(dict(carry=self._msg_carry(T)) if p.coupled else {}))
            return self._v_boot
        if self.save_acts:
            T = self.n_step
            p.step_policy(enc, self.H_all[:, T], self.C_all[:, T], done, self.h_fw, self.c_fw, self._pi_boot, action_scratch,
                          done_is_zero, mode=mode, u=u, seed=seed, env_id_base=env_id_base, step=step, step_dev=step_dev)
            return p.step_value(enc, self.h_fw, self.c_fw, done, self._h2, self._c2, action_scratch, self._v_boot,
                                done_is_zero)
        if p.fused_pv:
            p.step_policy_value(enc, self.h_fw, self.c_fw, done, self._pi_boot, action_scratch, self._v_boot, mode=mode, u=u,
                                seed=seed, env_id_base=env_id_base, step=step, step_dev=step_dev)
            return self._v_boot
        p.step_policy(enc, self.h_fw, self.c_fw, done, self.h_fw, self.c_fw, self._pi_boot, action_scratch,
                      done_is_zero, mode=mode, u=u, seed=seed, env_id_base=env_id_base, step=step, step_dev=step_dev)
        return p.step_value(enc, self.h_fw, self.c_fw, done, self._h2, self._c2, action_scratch, self._v_boot,
                            done_is_zero)

    def _loss_backward_fused(self, Hs):
        """`_loss(Hs).backward()` with the heads, the loss and the heads' backward as ONE pass over the h sequence
        (ops.heads_loss): this call is the root of the update's backward -- the head parameters receive their gradients here, the
        recurrence below Hs its dL/dh.  False: not available for this net (the caller takes the autograd chain)."""
        N, T, E = self.n_agent, self.n_step, self.E
        p = self.policy
        if Hs.dim() != 3 or not ops.heads_loss_supported(Hs, self.n_a, p.nbr_idx) or \
                (not self.identical_agent and not self.per_agent_optimizer) or Hs.stride(2) != 1 or Hs.stride(1) != Hs.shape[2]:
            return False
        prm = p.params
        action = self.buf_act.view(T * E, N)
        # the one-launch BPTT kernels take the heads' dL/dh as dy8 (32 bytes per row) and expand it themselves: no [N,T*E,64] tensor
        as_dy = self.save_acts and p.bptt_takes_head_dy and os.environ.get('NMARL_BPTT_HEAD_DY', '1') != '0'
        r = ops.heads_loss(Hs.detach(), prm['pi_w'], prm['pi_b'], prm['v_w'], prm['v_b'], action, p.nbr_idx, self.n_a,
                           self.Adv.view(N, T * E), self.R.view(N, T * E), self.v_coef, self.e_coef, want_dh=not as_dy)
        terms = r['terms']
        self.last_loss = (terms[:, 0], terms[:, 1], terms[:, 2], terms.sum(dim=1))
        for k in ('pi_w', 'pi_b', 'v_w', 'v_b'):
            prm[k].grad = r[k].reshape(prm[k].shape)
        Hs.backward(gradient=ops.head_dy_placeholder(r['dy8'], r['hw'], Hs) if as_dy else r['dh'])
        if as_dy and ops._pending_head_dy.pop(Hs.device, None) is not None:
            raise RuntimeError('the recurrence below Hs did not take the heads\' dy8 (ops.take_head_dy): its gradient is wrong')
        return True

    def _loss(self, Hs):
        """policies.py:20-30 / 232-255 with the batch mean taken over T*E."""
        N, T, E = self.n_agent, self.n_step, self.E
        p = self.policy
        # the critic's neighbour one-hots (policies.py:66-68) are gathered from the action bytes inside the op
        action = self.buf_act.view(T * E, N)
        logits, v = p.heads(Hs, action)                                       # [N,T*E,A], [N,T*E]
        adv = self.Adv.view(N, T * E)
        R = self.R.view(N, T * E)
        if not self.identical_agent and not self.per_agent_optimizer:
            # quirk Q6 (heterogeneous MA2C nets only, policies.py:241-254): the hetero branch builds prob_pi as
            # [N,1,T], so `prob_pi * ADV` broadcasts to [N,N,T] and every agent's log-probability is weighted by the
            # advantages of ALL agents: policy_loss = -sum_i mean_t(log pi_i[a] * sum_j ADV_j)
            adv = adv.sum(dim=0, keepdim=True).expand(N, T * E).contiguous()
        if ops.a2c_loss_supported(self.n_a):
            per_agent, terms = ops.a2c_loss(logits, v, action, adv, R, self.v_coef, self.e_coef)
            self.last_loss = (terms[:, 0], terms[:, 1], terms[:, 2], per_agent.detach())
            return per_agent.sum()
        pi = torch.softmax(logits, dim=-1)
        acts = action.t().long().unsqueeze(-1)                               # [N, T*E, 1]
        log_pi = torch.log(torch.clamp(pi, 1e-10, 1.0))
        entropy = -(pi * log_pi).sum(-1)
        logp_a = log_pi.gather(-1, acts).squeeze(-1)
        policy_loss = -(logp_a * adv).mean(-1)                               # [N]
        value_loss = (R - v).pow(2).mean(-1) * 0.5 * self.v_coef
        entropy_loss = -entropy.mean(-1) * self.e_coef
        per_agent = policy_loss + value_loss + entropy_loss
        self.last_loss = (policy_loss.detach(), value_loss.detach(), entropy_loss.detach(), per_agent.detach())
        return per_agent.sum()

    def update(self, R_end, rotate=True):
        """model.backward (models.py:34-42 / 211-215) for all replicas: R_end [N,E].  rotate=False: the caller hands the
        states / slot T of the rollout buffers over to the next batch itself (BatchedTrainer: ops.batch_epilogue).
        = the four phases below back to back; BatchedTrainer captures `update_grads` and `update_apply` in hipGraphs and
        replays them around the (eager) gradient exchange, calling the two host-only phases itself."""
        lr = self.update_begin()
        self.update_grads(R_end)
        self.update_reduce()
        self.update_apply(lr, rotate=rotate)
        self.update_end()

    def update_begin(self):
        """Host only: advance the lr schedule.  It counts lock-steps (environment steps per replica), the reference's
        get(n_step): the ini's total_step keeps its meaning for any number of replicas and ranks."""
        assert self.t == self.n_step, 'update() needs a full n_step batch (got %d)' % self.t
        self.cur_lr = self.lr_scheduler.get(self.n_step)
        return self.cur_lr

    @property
    def handoff_guarded(self):
        """This model launches in-launch hand-off kernels (one-launch coupled lock-step / BPTT): its optimiser step consults
        the device's hand-off status word and refuses a batch a timed-out wave contributed to (ops.rmsprop_tf_clip)."""
        return bool(self.policy.coupled) and self.device.type == 'cuda'

    def update_grads(self, R_end):
        """Device work of one update up to the flat gradient: returns / advantages, loss, backward (no host
        synchronisation, static shapes and addresses: capturable)."""
        alpha = self.coop_gamma if self.coop_gamma >= 0 else -1.0
        T = self.n_step
        if self.save_acts and self.policy.pv_one_launch(self.E):
            # the critic's neighbour-action term of all T lock-steps in one launch (policies.py:59-77), then the values
            # in the [T,N,E] order the return scan reads
            with torch.no_grad():
                vn = self.buf_vn.view(self.n_agent, T * self.E)
                ops.nbr_action_value(self.buf_act.view(T * self.E, self.n_agent), self.policy.nbr_idx,
                                     self.policy.params['v_w'][:, self.n_lstm:], self.n_a, out=vn, accumulate=True)
                self.buf_v.copy_(self.buf_vn.permute(1, 0, 2))
        ops.nstep_return(self.buf_r, self.buf_v, self.buf_done_post, R_end.contiguous(), self.gamma, alpha,
                         self.dist_dev, self.R, self.Adv)
        ps = self.policy.params
        ps.begin_backward()
        FP = self.buf_fp[:T].permute(1, 0, 2, 3).reshape(self.n_agent, T * self.E, self.n_a)
        X = self.buf_x[:T]            # compact [T,E,N,n_feat] slab: the encoders' kernels gather the neighbours themselves
        if self.save_acts:
            kw = {}
            if self.S_bits is not None and self.policy._bits_steps == T:      # every lock-step of this batch went through the kernel that writes them
                kw['S_bits'] = self.S_bits
            Hs = self.policy.unroll_saved(X, FP, self.S_buf, self.G_buf, self.H_all, self.C_all,
                                          self.buf_done_pre, masked_steps=self.masked_steps, S_ext=self.S_ext, **kw)
        else:
            Hs = self.policy.unroll(X, FP, self.buf_done_pre, self.h_bw, self.c_bw, masked_steps=self.masked_steps)
        if not self._loss_backward_fused(Hs):
            loss = self._loss(Hs)
            loss.backward()
        ps.end_backward()
        if ps.mask is not None:          # entries of variables the reference does not create (heterogeneous nets)
            ps.grad.mul_(ps.mask)
        if self._status_on_wire:
            # a rank whose in-launch hand-off timed out contributes invalid gradients: every rank must refuse the step (and recover
            # in lock-step, BatchedTrainer).  The status word rides in the float behind the gradient, inside the ONE all-reduce
            ps.grad_tail.copy_(ops.handoff_status(self.device)[:1])           # (int32 -> f32: an element-wise kernel)

    @property
    def _status_on_wire(self):
        return self.dist_group is not None and self.handoff_guarded and self.save_acts

    def update_reduce(self):
        """The data-parallel exchange: ONE flat all-reduce per update (RCCL over xGMI) -- the gradient and, behind it, one float
        that is non-zero iff some rank's in-launch hand-off timed out in this batch (the reference has no collective)."""
        if self.dist_group is None:
            return
        import torch.distributed as dist
        dist.all_reduce(self.policy.params.grad_wire, group=self.dist_group)
        self.allreduce_calls = getattr(self, 'allreduce_calls', 0) + 1

    def update_apply(self, lr, rotate=True, lr_dev=None):
        """clip_by_global_norm + RMSProp on the flat buffers (policies.py:32-39, 257-264); lr_dev: device scalar that
        overrides `lr` (captured updates: the host refreshes it when the schedule moves)."""
        ps = self.policy.params
        scale = 1.0 / self.world_size
        guard = self.handoff_guarded
        if self._status_on_wire:             # some rank timed out (summed tail != 0): this rank's status word is raised as well
            st = ops.handoff_status(self.device)[:1]
            torch.maximum(st, (ps.grad_tail != 0).to(torch.int32), out=st)
        if self.per_agent_optimizer:
            ops.rmsprop_tf_clip(ps.flat, ps.grad, ps.ms, ps.scratch, lr, self.rmsp_alpha, self.rmsp_epsilon,
                                self.max_grad_norm, scale, self.grad_norm, lr_dev=lr_dev, guard=guard)
        else:
            n = self.n_agent * ps.P
            ops.rmsprop_tf_clip(ps.flat.view(1, n), ps.grad.view(1, n), ps.ms.view(1, n), ps.scratch, lr,
                                self.rmsp_alpha, self.rmsp_epsilon, self.max_grad_norm, scale, self.grad_norm,
                                lr_dev=lr_dev, guard=guard)
        self._after_apply()
        if rotate:
            T = self.n_step
            # states_bw <- states_fw (policies.py:115, 211)
            self.h_bw.copy_(self.h_fw)
            self.c_bw.copy_(self.c_fw)
            # slot T (bootstrap inputs) is slot 0 of the next batch
            self.buf_x[0].copy_(self.buf_x[T])
            self.buf_fp[0].copy_(self.buf_fp[T])

    def _after_apply(self):
        """Hook behind the optimiser step (ConseNet: the consensus averaging)."""

    def update_end(self):
        """Host only: the batch is consumed."""
        self.t = 0
        self.policy._enc_was_saved = False       # the saved encoder outputs belonged to this batch (the next rollout sets it again)
        self.policy._bits_steps = 0
        self.policy._mm_was_saved = False

    # ------------------------------------------------------------------ reference API (E = 1)
    def _obs_to_slab(self, obs):
        """list of N 1-D arrays (reference env) -> [1,N,n_obs] slab (+ fingerprints for ia2c_fp).  The slab has one
        n_feat-wide slot per (own, neighbour 1, ..): agents with narrower observations are zero padded inside their
        slots (heterogeneous systems), the reference's tightly concatenated vectors are scattered accordingly."""
        p = self.policy
        F, A = self.n_feat, self.n_a
        slab = np.zeros((1, self.n_agent, p.n_obs), dtype=np.float32)
        fp = None
        if not self._is_ma2c() and self.uses_fingerprint_obs():
            fp = np.zeros((self.n_agent, 1, p.n_na), dtype=np.float32)
        for i in range(self.n_agent):
            o = np.asarray(obs[i], dtype=np.float32)
            slab[0, i, :p.n_own[i]] = o[:p.n_own[i]]
            if self._is_ma2c():
                continue
            pos = p.n_own[i]
            for j in p.obs_order[i]:                     # the env's concatenation order -> the slab's ascending slots
                k = p.nbrs[i].index(j)
                slab[0, i, (k + 1) * F:(k + 1) * F + p.n_own[j]] = o[pos:pos + p.n_own[j]]
                pos += p.n_own[j]
            if fp is not None:
                for j in p.obs_order[i]:
                    k = p.nbrs[i].index(j)
                    fp[i, 0, k * A:k * A + p.n_a_ls[j]] = o[pos:pos + p.n_a_ls[j]]
                    pos += p.n_a_ls[j]
        slab = torch.from_numpy(slab).to(self.device)
        if self._is_ma2c():
            # MA2C envs hand over the own features only; neighbours are gathered here
            x = slab[:, :, :F].transpose(0, 1).contiguous()                   # [N,1,F]
            slab = torch.cat([x, ops.nbr_gather(x, p.nbr_idx)], dim=-1).transpose(0, 1).contiguous()
        return slab, fp

    def uses_fingerprint_obs(self):
        return False

    def _done_t(self, done):
        return torch.full((self.E,), float(bool(done)), dtype=F32, device=self.device)

    def _na_onehot_from_list(self, nactions):
        p = self.policy
        na = np.zeros((self.n_agent, 1, p.n_na), dtype=np.float32)
        for i in range(self.n_agent):
            for k, a in enumerate(np.asarray(nactions[i]).reshape(-1)):
                na[i, 0, k * self.n_a + int(a)] = 1.0
        return torch.from_numpy(na).to(self.device)

    def forward(self, obs, done, nactions=None, out_type='p'):
        """IA2C.forward (models.py:44-51): list of N pi arrays ('p') or N scalars ('v')."""
        slab, _ = self._obs_to_slab(obs)
        self._cur_slab = slab
        d = self._done_t(done)
        if out_type.startswith('p'):
            pi = self._policy_step(slab, d)
            return [x[:self.n_a_ls[i]] for i, x in enumerate(pi[:, 0].cpu().numpy())]
        v = self._value_step(slab, d, self._na_onehot_from_list(nactions))
        return [x for x in v[:, 0].cpu().numpy()]

    def add_transition(self, ob, naction, action, reward, value, done):
        """models.py:26-32 -> device buffers (slot t).  `naction` is not stored: it is
        env.get_neighbor_action(action) (utils.py:172, cacc_env.py:125-129), i.e. a function of `action`, and
        update() rebuilds the critic's one-hots from the action bytes."""
        t = self.t
        slab, fp = self._obs_to_slab(ob)
        self.buf_x[t].copy_(slab)
        self.buf_act[t].copy_(torch.as_tensor(np.asarray(action, dtype=np.uint8).reshape(1, -1)))
        self.buf_v[t].copy_(torch.as_tensor(np.asarray(value, dtype=np.float32).reshape(-1, 1)))
        self.buf_done_pre[t].fill_(float(self._prev_done))
        self._prev_done = bool(done)
        r = torch.as_tensor(np.asarray(reward, dtype=np.float32)).to(self.device)
        self.record(r.view(self.buf_r[t].shape), torch.full((self.E,), int(bool(done)), dtype=torch.uint8,
                                                           device=self.device))

    def backward(self, Rends, dt=0, summary_writer=None, global_step=None):
        R_end = torch.as_tensor(np.asarray(Rends, dtype=np.float32).reshape(self.n_agent, 1)).to(self.device)
        self.update(R_end)
        if self.handoff_guarded:
            # the reference API has no trainer that re-runs a batch: a timed-out in-launch hand-off (whose optimiser step the
            # device refused) is an error here, not a silent no-op (this path synchronises every lock-step anyway)
            ops.check_coupled_status(self.device)

    def reset(self):
        self.reset_states()
        self._prev_done = True          # OnPolicyBuffer keeps the done BEFORE each step (agents/utils.py:732-739)

    _prev_done = True

    # ------------------------------------------------------------------ checkpoints (models.py:53-82)
    def save(self, model_dir, global_step):
        """`checkpoint-<step>.pt` (the reference writes TF Saver files `checkpoint-<step>.*`, models.py:53-58; the two
        formats are not interchangeable): a dict of plain tensors / numbers only, so it loads with
        torch.load(weights_only=True) -- variables under the reference's names and ragged shapes, the RMSProp slots,
        the lr schedule position."""
        path = model_dir + 'checkpoint-%d.pt' % int(global_step)
        ps = self.policy.params
        sched = getattr(self, 'lr_scheduler', None)
        torch.save({'variables': {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ps.ref_variables()},
                    'rmsprop_ms': ps.ms.detach().cpu(), 'name': self.name, 'global_step': int(global_step),
                    'lr_n': float(sched.n) if sched is not None else -1.0}, path)
        # keep the 5 newest, like tf.train.Saver(max_to_keep=5)
        found = sorted(self._list_checkpoints(model_dir))
        for step, f in found[:-5]:
            os.remove(os.path.join(model_dir, f))

    @staticmethod
    def _list_checkpoints(model_dir):
        out = []
        for f in os.listdir(model_dir):
            if f.startswith('checkpoint'):
                tokens = f.split('.')[0].split('-')
                if len(tokens) == 2 and tokens[1].isdigit():
                    out.append((int(tokens[1]), f))
        return out

    def load(self, model_dir, checkpoint=None):
        """models.py:60-82: newest `checkpoint-<step>` of the directory, or the given step; False if there is none."""
        save_file = None
        if os.path.exists(model_dir):
            if checkpoint is None:
                found = sorted(self._list_checkpoints(model_dir))
                if found:
                    save_file = found[-1][1]
            else:
                save_file = 'checkpoint-%d.pt' % int(checkpoint)
        if save_file is not None and os.path.isfile(os.path.join(model_dir, save_file)):
            blob = torch.load(os.path.join(model_dir, save_file), weights_only=True, map_location='cpu')
            self.policy.params.load_ref_variables({k: v.numpy() for k, v in blob['variables'].items()})
            if 'rmsprop_ms' in blob:
                self.policy.params.ms.copy_(blob['rmsprop_ms'])
            if blob.get('lr_n', -1.0) >= 0 and getattr(self, 'lr_scheduler', None) is not None:
                self.lr_scheduler.n = blob['lr_n']       # resume the lr decay where it stopped
            logging.info('Checkpoint loaded: %s' % save_file)
            return True
        logging.error('Can not find old checkpoint for %s' % model_dir)
        return False


class IA2C_FP(IA2C):
    """Fingerprint IA2C (models.py:161-188): neighbour policies appended to the observation."""
    policy_cls = FPPolicy
    name = 'ia2c_fp'

    def uses_fingerprint_obs(self):
        return True

    def forward(self, obs, done, nactions=None, out_type='p'):
        # the reference env delivers the neighbours' fingerprints inside `obs`; scatter them back
        # into the per-agent fingerprint table the batched policy gathers from
        _, fpg = self._obs_to_slab(obs)
        self.fp.copy_(self._ungather_fp(fpg))
        return super().forward(obs, done, nactions, out_type)

    def _ungather_fp(self, fpg):
        """[N,1,m_max*A] gathered neighbour fingerprints -> [N,1,A] table (every agent's
        fingerprint appears in at least one neighbour's slot)."""
        p = self.policy
        tab = self.fp_uniform.cpu().numpy().copy()
        idx = p.nbr_idx.cpu().numpy()
        for i in range(self.n_agent):
            for k in range(p.m_max):
                j = idx[i, k]
                if j >= 0:
                    tab[j, 0] = fpg[i, 0, k * self.n_a:(k + 1) * self.n_a]
        return torch.from_numpy(tab).to(self.device)

    def add_transition(self, ob, naction, action, reward, value, done):
        _, fpg = self._obs_to_slab(ob)
        self.buf_fp[self.t].copy_(self._ungather_fp(fpg))
        super().add_transition(ob, naction, action, reward, value, done)


class MA2C_NC(IA2C):
    """NeurComm (models.py:191-258): one optimiser over the whole meta-DNN."""
    policy_cls = NCMultiAgentPolicy
    per_agent_optimizer = False
    name = 'ma2c_nc'

    def forward(self, obs, done, ps, actions=None, out_type='p'):
        """MA2C_NC.forward (models.py:217-224): [N,A] ('p') or [N] ('v')."""
        slab, _ = self._obs_to_slab(obs)
        self.fp.copy_(self._pad_policies(ps))
        d = self._done_t(done)
        if out_type.startswith('p'):
            return self._unpad_policies(self._policy_step(slab, d)[:, 0].cpu().numpy())
        a = torch.as_tensor(np.asarray(actions, dtype=np.uint8).reshape(1, -1)).to(self.device)
        na = ops.nbr_onehot(a, self.policy.nbr_idx, self.n_a)
        return self._value_step(slab, d, na)[:, 0].cpu().numpy()

    def add_transition(self, ob, p, action, reward, value, done):
        t = self.t
        slab, _ = self._obs_to_slab(ob)
        self.buf_x[t].copy_(slab)
        self.buf_fp[t].copy_(self._pad_policies(p))
        a = torch.as_tensor(np.asarray(action, dtype=np.uint8).reshape(1, -1)).to(self.device)
        self.buf_act[t].copy_(a)
        self.buf_v[t].copy_(torch.as_tensor(np.asarray(value, dtype=np.float32).reshape(-1, 1)))
        self.buf_done_pre[t].fill_(float(self._prev_done))
        self._prev_done = bool(done)
        r = torch.as_tensor(np.asarray(reward, dtype=np.float32)).to(self.device)
        self.record(r.view(self.buf_r[t].shape), torch.full((self.E,), int(bool(done)), dtype=torch.uint8,
                                                           device=self.device))


class MA2C_IC3(MA2C_NC):
    """CommNet (models.py:278-292)."""
    policy_cls = IC3MultiAgentPolicy
    name = 'ma2c_ic3'


class IA2C_CU(MA2C_NC):
    """ConseNet (models.py:261-275): MA2C-style single optimiser + consensus averaging of the LSTM weights
    over each neighbourhood after every update (policies.py:351-364)."""
    policy_cls = ConsensusPolicy
    name = 'ma2c_cu'

    def _after_apply(self):
        self.policy.consensus_update()


class MA2C_DIAL(MA2C_NC):
    """DIAL (models.py:295-309)."""
    policy_cls = DIALMultiAgentPolicy
    name = 'ma2c_dial'
