"""TEST-ONLY: run the product's HOST logic on the CPU by patching the HIP-op
wrappers of `deeprl_network_amd.ops` with the torch-CPU restatements of
oracle/ops_ref.py.  This is how `-m "not gpu"` tests cover the Python side
(buffer indexing, quirk ordering, parameter layout, checkpoint naming) in a
container without a GPU.  The product itself has no CPU path: outside this
context manager every op raises on CPU tensors.
"""
import contextlib

from deeprl_network_amd import ops
from oracle import ops_ref

_NAMES = ['nbr_gather', 'nbr_mean', 'nbr_gather_bwd', 'nbr_mean_bwd', 'cell_bwd', 'nbr_onehot', 'lstm_cell', 'lstm_cell_infer', 'lstm_sequence', 'lstm_sequence_x', 'lstm_sequence_saved', 'bptt_supported', 'lstm_bptt_wimage', 'bptt_step', 'bptt_step_db_parts', 'bptt_seq', 'bptt_coupled', 'bptt_coupled_supported', 'lstm_bptt_msg_wimage', 'reverse_neighbor_table', 'dial_adjoint_supported', 'dial_adjoint_images', 'dial_msg_adjoint', 'dial_adjoint_bias_parts', 'lstm_wimage', 'lstm_msg_wimage', 'msg_supported', 'ob_encoder_supported', 'lstm_ob_wimage', 'step_sync_words', 'step_handoff_supported', 'xside_supported', 'lstm_step_fused', 'lstm_step_policy', 'lstm_step_value', 'lstm_step_policy_value', 'step_enc_supported', 'step_enc1_supported', 'step_enc_spec', 'bias_act_', 'fc_fwd', 'fc_fwd_multi', 'onehot_argmax_add_', 'fc_bwd', 'fc_concat', 'fc_supported', 'thin_linear', 'thin_linear_bwd', 'nbr_action_value', 'nbr_action_value_bwd', 'heads', 'heads_supported', 'wgrad', 'linear', 'a2c_loss', 'a2c_loss_supported', 'sample_actions',
          'nstep_return', 'rmsprop_tf_clip', 'batch_epilogue']


@contextlib.contextmanager
def cpu_ops():
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, getattr(ops_ref, n))
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)


class CpuCaccBatchEnv:
    """TEST-ONLY stand-in for envs.cacc_env.CACCBatchEnv on CPU tensors, built on the oracle
    (oracle/cacc_ref.py fp32 + oracle/philox.py): same attributes / reset / step contract."""

    def __init__(self, config, num_envs=1, device='cpu', env_id_base=0, seed=None):
        import numpy as np
        import torch
        from oracle.cacc_ref import CaccBatchRef, CaccParams
        self.config, self.E, self.device = config, num_envs, torch.device(device)
        self.p = CaccParams(config=config)
        self.ref = CaccBatchRef(self.p, E=num_envs, dtype=np.float32)
        self.n_agent, self.agent, self.name = self.p.n_agent, self.p.agent, self.p.name
        self.coop_gamma, self.T, self.batch_size = self.p.coop_gamma, self.p.T, self.p.batch_size
        self.seed = self.p.seed if seed is None else seed
        self.env_id_base = env_id_base
        self.neighbor_mask, self.distance_mask = self.ref.neighbor_mask, self.ref.distance_mask
        self.n_a, self.n_a_ls = 4, [4] * self.n_agent
        self.n_s_ls = [5 if self.agent.startswith('ma2c') else 5 * (1 + int(self.neighbor_mask[i].sum()))
                       for i in range(self.n_agent)]
        self.episode = torch.zeros(num_envs, dtype=torch.int32)
        self.obs = torch.zeros(num_envs, self.n_agent, 15)
        self.done = torch.zeros(num_envs, dtype=torch.uint8)
        self.reward = torch.zeros(num_envs)
        self.global_reward = torch.zeros(num_envs)

    train_mode = property(lambda self: self.ref.train_mode, lambda self, f: setattr(self.ref, 'train_mode', bool(f)))
    compact_obs = False

    def set_compact_obs(self, flag=True):
        import torch
        self.compact_obs = bool(flag)
        self.obs = torch.zeros(self.E, self.n_agent, 5 if flag else 15)
        return True

    def _emit(self):
        import torch
        from oracle.cacc_ref import gather_line
        vs = self.ref.veh_state()
        self.obs.copy_(torch.from_numpy(np_f32(vs) if self.compact_obs else gather_line(vs)))
        return self.obs

    def reset(self, mask=None, u0=None):
        import numpy as np
        from oracle import philox
        m = np.ones(self.E, bool) if mask is None else mask.numpy().astype(bool)
        ep = self.episode.numpy()
        U = philox.reset_uniform(self.seed, self.env_id_base + np.arange(self.E), ep) if u0 is None else u0.numpy()
        if u0 is None:
            ep[m] += 1
        self.ref.reset(U, mask=None if mask is None and not hasattr(self.ref, 'h') else m)
        return self._emit()

    def step(self, action, auto_reset=False, obs_out=None, reward_out=None, done_out=None, greward_out=None):
        import torch
        _, r, d, g = self.ref.step(action.numpy())
        reward = self.reward if reward_out is None else reward_out
        greward = self.global_reward if greward_out is None else greward_out
        done = self.done if done_out is None else done_out
        reward.copy_(torch.from_numpy(np_f32(r)))
        greward.copy_(torch.from_numpy(np_f32(g)))
        done.copy_(torch.from_numpy(d.astype('uint8')))
        if auto_reset and d.any():
            self.reset(mask=torch.from_numpy(d.astype('uint8')))
        else:
            self._emit()
        if obs_out is not None:
            obs_out.copy_(self.obs)
        return (self.obs if obs_out is None else obs_out), reward, done, greward


def np_f32(x):
    import numpy as np
    return np.asarray(x, dtype=np.float32)


class CpuRealNetBatchEnv:
    """TEST-ONLY stand-in for envs.real_net_env.RealNetBatchEnv on CPU tensors, built on the oracle
    (oracle/realnet_ref.py fp32 + oracle/philox.py): same attributes / reset / step contract, heterogeneous agents."""

    def __init__(self, config, num_envs=1, device='cpu', env_id_base=0, seed=None):
        import numpy as np
        import torch
        from oracle.realnet_ref import TOPO, NetBatchRef, NetParams
        self.config, self.E, self.device = config, num_envs, torch.device(device)
        self.p = NetParams(config=config)
        self.ref = NetBatchRef(self.p, E=num_envs, dtype=np.float32)
        tp = self.topo = TOPO
        self.name, self.agent, self.coop_gamma, self.T = config.get('scenario'), self.p.agent, self.p.coop_gamma, self.p.T
        self.seed = self.p.seed if seed is None else seed
        self.env_id_base = env_id_base
        self.n_agent, self.n_a, self.n_a_ls = tp.N, tp.A, list(tp.n_a_ls)
        self.n_feat, self.n_feat_ls = tp.L, list(tp.n_s_ls)
        self.neighbor_mask, self.distance_mask = tp.neighbor_mask, tp.distance_mask
        self.n_s_ls = list(tp.n_s_ls) if self.agent.startswith('ma2c') else \
            [tp.n_s_ls[i] + sum(tp.n_s_ls[j] for j in tp.nbrs[i]) for i in range(tp.N)]
        self.batch_size = None
        self.train_mode = True
        self.episode = torch.zeros(num_envs, dtype=torch.int32)
        self.obs = torch.zeros(num_envs, tp.N, tp.L * (1 + tp.m_max))
        self.done = torch.zeros(num_envs, dtype=torch.uint8)
        self.reward = torch.zeros((num_envs, tp.N) if self.coop_gamma >= 0 else (num_envs,))
        self.global_reward = torch.zeros(num_envs)

    def state_tensors(self):
        return [self.obs, self.episode, self.done]

    def _emit(self):
        import torch
        from oracle.realnet_ref import gather_net
        self.obs.copy_(torch.from_numpy(gather_net(self.ref.obs())))
        return self.obs

    def reset(self, mask=None, u0=None):
        import numpy as np
        from oracle import philox
        m = np.ones(self.E, bool) if mask is None else mask.numpy().astype(bool)
        ep = self.episode.numpy()
        if u0 is None:
            w = philox.philox4x32(np.asarray(self.env_id_base + np.arange(self.E), dtype=np.uint64), 0,
                                  np.asarray(ep, dtype=np.uint64), 0, self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)
            U = philox.u01(np.stack(w, axis=-1))                 # the four demand-scale words, stream RESET
            ep[m] += 1
        else:
            U = u0.numpy()
        self.ref.reset(0.8 + 0.4 * np.asarray(U, dtype=np.float32), mask=None if mask is None and not hasattr(self.ref, 'q') else m)
        return self._emit()

    def step(self, action, auto_reset=False, obs_out=None, reward_out=None, done_out=None, greward_out=None):
        import torch
        _, r, d, g = self.ref.step(action.numpy())
        reward = self.reward if reward_out is None else reward_out
        greward = self.global_reward if greward_out is None else greward_out
        done = self.done if done_out is None else done_out
        reward.copy_(torch.from_numpy(np_f32(r)))
        greward.copy_(torch.from_numpy(np_f32(g)))
        done.copy_(torch.from_numpy(d.astype('uint8')))
        if auto_reset and d.any():
            self.reset(mask=torch.from_numpy(d.astype('uint8')))
        else:
            self._emit()
        if obs_out is not None:
            obs_out.copy_(self.obs)
        return (self.obs if obs_out is None else obs_out), reward, done, greward
