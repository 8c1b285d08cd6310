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

_NAMES = ['nbr_gather', 'nbr_mean', 'nbr_gather_bwd', 'nbr_mean_bwd', 'cell_bwd', 'nbr_onehot', 'lstm_cell', 'lstm_cell_infer', 'lstm_sequence', 'lstm_step_fused', 'lstm_step_policy', 'lstm_step_value', 'bias_act_', 'fc_fwd', 'fc_fwd_multi', 'fc_bwd', 'fc_concat', 'fc_supported', 'thin_linear', 'thin_linear_bwd', 'nbr_action_value', 'nbr_action_value_bwd', 'heads', 'heads_supported', 'wgrad', 'linear', 'a2c_loss', 'a2c_loss_supported', 'sample_actions',
          'nstep_return', 'rmsprop_tf_clip']


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

    def _emit(self):
        import torch
        from oracle.cacc_ref import gather_line
        self.obs.copy_(torch.from_numpy(gather_line(self.ref.veh_state())))
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
