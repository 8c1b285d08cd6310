"""CACC platoon environment on MI355X: E lock-stepped replicas stepped by the HIP
kernels of csrc/cacc.hip through the C-ABI (include/nmarl.h).

Host-side mirror of the reference's envs/cacc_env.py:
  * `CACCBatchEnv` -- the batched, device-resident environment ([E,N,...] tensors);
  * `CACCEnv`      -- the reference's own duck-type (cacc_env.py:13-343: list-of-
                      arrays observations, scalar reward, global `np.random`
                      seeding) as an E=1 adapter, so the reference's Trainer
                      semantics and ini files keep working unchanged.

All arithmetic happens in the HIP kernels; there is no CPU fallback.
"""
import ctypes
import logging

import numpy as np
import pandas as pd
import torch

from .. import _lib

N_FEAT = 5
N_OBS = 15


def _params_from_config(config, train_mode=True):
    """ENV_CONFIG section -> nmarl_cacc_params_t; keys of cacc_env.py:320-343."""
    p = _lib.CaccParams()
    p.dt = config.getfloat('control_interval_sec')
    p.T = int(config.getint('episode_length_sec') / config.getfloat('control_interval_sec'))
    p.batch_size = config.getint('batch_size')
    p.h_min = config.getfloat('headway_min')
    p.h_star = config.getfloat('headway_target')
    p.h_s = config.getfloat('headway_st')
    p.h_g = config.getfloat('headway_go')
    p.v_max = config.getfloat('speed_max')
    p.v_star = config.getfloat('speed_target')
    p.u_min = config.getfloat('accel_min')
    p.u_max = config.getfloat('accel_max')
    p.reward_a = config.getfloat('reward_v')
    p.reward_b = config.getfloat('reward_u')
    p.G = config.getfloat('collision_penalty')
    name = config.get('scenario').split('_')[1]
    if not (name.startswith('catchup') or name.startswith('slowdown')):
        raise ValueError('unknown CACC scenario %r' % config.get('scenario'))
    p.scenario = 0 if name.startswith('catchup') else 1
    p.train_mode = 1 if train_mode else 0
    p.per_agent_reward = 0 if config.getfloat('coop_gamma') < 0 else 1
    return p, name


def line_graph(n):
    """neighbor_mask / distance_mask of cacc_env.py:253-268."""
    idx = np.arange(n)
    dist = np.abs(idx[:, None] - idx[None, :]).astype(int)
    return (dist == 1).astype(int), dist


class CACCBatchEnv:
    """E independent platoons stepped in lock-step on one GPU.

    State (HBM, fp32 SoA): h, v, u [E,8]; t [E] i32; collided [E] u8;
    v0_init [E]; observation slab obs [E,8,15] (own + 2 neighbour slots).
    `env_id_base` makes Philox streams global across data-parallel ranks.
    """

    def __init__(self, config, num_envs=1, device='cuda', env_id_base=0, seed=None):
        self.config = config
        self.E = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.NmarlError('CACCBatchEnv needs a HIP device; there is no CPU path')
        self.params, self.name = _params_from_config(config)
        self.n_agent = config.getint('n_vehicle')
        if self.n_agent != 8:
            raise _lib.NmarlError('the gfx950 CACC kernel maps one platoon to an 8-lane group: n_vehicle must be 8')
        self.agent = config.get('agent')
        self.coop_gamma = config.getfloat('coop_gamma')
        self.T = self.params.T
        self.batch_size = self.params.batch_size
        self.dt = self.params.dt
        self.seed = config.getint('seed') if seed is None else int(seed)
        self.env_id_base = int(env_id_base)
        self.n_a = 4
        self.n_a_ls = [4] * self.n_agent
        self.neighbor_mask, self.distance_mask = line_graph(self.n_agent)
        self.n_s_ls = [N_FEAT if self.agent.startswith('ma2c') else N_FEAT * (1 + int(self.neighbor_mask[i].sum()))
                       for i in range(self.n_agent)]
        self._train_mode = True
        E, N, d = self.E, self.n_agent, self.device
        f32 = dict(dtype=torch.float32, device=d)
        self.h = torch.zeros(E, N, **f32)
        self.v = torch.zeros(E, N, **f32)
        self.u = torch.zeros(E, N, **f32)
        self.t = torch.zeros(E, dtype=torch.int32, device=d)
        self.collided = torch.zeros(E, dtype=torch.uint8, device=d)
        self.v0_init = torch.zeros(E, **f32)
        self.obs = torch.zeros(E, N, N_OBS, **f32)
        self.reward = torch.zeros((E, N) if self.params.per_agent_reward else (E,), **f32)
        self.done = torch.zeros(E, dtype=torch.uint8, device=d)
        self.global_reward = torch.zeros(E, **f32)
        self.episode = torch.zeros(E, dtype=torch.int32, device=d)
        self.fp = torch.full((E, N, self.n_a), 1.0 / self.n_a, **f32)

    def state_tensors(self):
        """Everything a step mutates (snapshot / restore around hipGraph capture)."""
        return [self.h, self.v, self.u, self.t, self.collided, self.v0_init, self.obs, self.episode, self.done]

    compact_obs = False

    def set_compact_obs(self, flag=True):
        """Compact observation [E,8,5] (each vehicle's own features, SURVEY.md 8d's 347-B layout) instead of the
        pre-gathered [E,8,15]: the consumer gathers the neighbours (agents/policies.py `_ob_part`).  Batched engine only
        (the E = 1 reference duck-type keeps the reference's concatenated observations)."""
        self.compact_obs = bool(flag)
        self.params.compact_obs = 1 if flag else 0
        self.obs = torch.zeros(self.E, self.n_agent, N_FEAT if flag else N_OBS, dtype=torch.float32, device=self.device)
        return True

    @property
    def train_mode(self):
        return self._train_mode

    @train_mode.setter
    def train_mode(self, flag):
        self._train_mode = bool(flag)
        self.params.train_mode = 1 if flag else 0

    def reset(self, mask=None, u0=None):
        """Reset all replicas (or those with mask != 0).  `u0` [E] fp32 supplies
        the initial-condition uniforms (legacy / test seeds); otherwise they come
        from Philox(seed, env_id, episode) inside the kernel."""
        P = _lib.ptr
        rc = _lib.lib.nmarl_cacc_reset(
            ctypes.byref(self.params), self.E, P(mask, torch.uint8), P(u0, torch.float32),
            self.seed, self.env_id_base, P(self.episode), P(self.h), P(self.v), P(self.u), P(self.t),
            P(self.collided), P(self.v0_init), P(self.obs), P(self.fp), self.n_a, _lib.stream())
        _lib.check(rc, 'nmarl_cacc_reset')
        return self.obs

    supports_fused_encode = True

    def inkernel_step(self, auto_reset=False, obs_out=None, reward_out=None, done_out=None, greward_out=None):
        """Arguments of `step` for the policy's lock-step launch to run the env step itself behind its action draw
        (csrc/lstm_mfma.hip ENV: ops.step_enc_spec(env=...)): same state tensors, same outputs, the actions are the launch's own."""
        if not self.compact_obs:
            raise _lib.NmarlError('the in-launch env step writes the compact observation')
        return dict(params=self.params, h=self.h, v=self.v, u=self.u, t=self.t, collided=self.collided, v0_init=self.v0_init,
                    obs_out=self.obs if obs_out is None else obs_out, reward=self.reward if reward_out is None else reward_out,
                    done=self.done if done_out is None else done_out,
                    global_reward=self.global_reward if greward_out is None else greward_out, auto_reset=bool(auto_reset),
                    seed=self.seed, env_id_base=self.env_id_base, episode=self.episode)

    def step(self, action, auto_reset=False, obs_out=None, reward_out=None, done_out=None, greward_out=None, encode=None):
        """action [E,8] uint8 -> (obs [E,8,15], reward [E]|[E,8], done [E] u8, global_reward [E]).
        By default the results land in this env's persistent buffers (overwritten every step); the
        `*_out` tensors redirect them, e.g. straight into slot t of the trainer's rollout buffers.
        encode (compact observation only): dict(w_ob, b_ob[, w_fp, b_fp, fp], nbr_idx, out, act) -- the NEXT lock-step's input
        encoders run in the same launch on the observation this step produces (nmarl_cacc_step_encode)."""
        P = _lib.ptr
        obs = self.obs if obs_out is None else obs_out
        reward = self.reward if reward_out is None else reward_out
        done = self.done if done_out is None else done_out
        greward = self.global_reward if greward_out is None else greward_out
        if encode is not None:
            en = _lib.CaccEncode()
            S = P
            w, b, out = encode['w_ob'], encode['b_ob'], encode['out']
            if w.shape[1:] != (15, 64) or w.stride(2) != 1 or w.stride(1) != 64 or out.stride(2) != 1:
                raise _lib.NmarlError('fused encode: w_ob must be [N,15,64] panels, out [N,E,>=128] with unit column stride')
            en.w_ob, en.w_ob_sn, en.b_ob, en.b_ob_sn = S(w, strided=True), w.stride(0), S(b, strided=True), b.stride(0)
            en.n_parts, en.act = (2 if encode.get('w_fp') is not None else 1), int(encode['act'])
            if en.n_parts == 2:
                wf, bf, fp = encode['w_fp'], encode['b_fp'], encode['fp']
                if wf.shape[1:] != (8, 64) or wf.stride(2) != 1 or wf.stride(1) != 64 or fp.shape[1:] != (self.E, 4) or not fp[0].is_contiguous():
                    raise _lib.NmarlError('fused encode: w_fp must be [N,8,64] panels, fp [N,E,4]')
                en.w_fp, en.w_fp_sn, en.b_fp, en.b_fp_sn = S(wf, strided=True), wf.stride(0), S(bf, strided=True), bf.stride(0)
                en.fp, en.fp_sn = S(fp, strided=True), fp.stride(0)
            en.nbr_idx = P(encode['nbr_idx'], torch.int32)
            en.out, en.out_sn, en.out_row = S(out, strided=True), out.stride(0), out.stride(1)
            rc = _lib.lib.nmarl_cacc_step_encode(
                ctypes.byref(self.params), self.E, P(action, torch.uint8), P(self.h), P(self.v), P(self.u),
                P(self.t), P(self.collided), P(self.v0_init), P(obs, torch.float32), P(reward, torch.float32),
                P(done, torch.uint8), P(greward, torch.float32), 1 if auto_reset else 0, self.seed,
                self.env_id_base, P(self.episode), ctypes.byref(en), _lib.stream())
            _lib.check(rc, 'nmarl_cacc_step_encode')
            return obs, reward, done, greward
        rc = _lib.lib.nmarl_cacc_step(
            ctypes.byref(self.params), self.E, P(action, torch.uint8), P(self.h), P(self.v), P(self.u),
            P(self.t), P(self.collided), P(self.v0_init), P(obs, torch.float32), P(reward, torch.float32),
            P(done, torch.uint8), P(greward, torch.float32), 1 if auto_reset else 0, self.seed,
            self.env_id_base, P(self.episode), _lib.stream())
        _lib.check(rc, 'nmarl_cacc_step')
        return obs, reward, done, greward

    def update_fingerprint(self, fp):
        self.fp = fp

    def get_fingerprint(self):
        return self.fp


class CACCEnv:
    """Drop-in for the reference `CACCEnv` (cacc_env.py:13-343): same constructor
    (`config['ENV_CONFIG']`), attributes and methods, one replica, observations
    as a list of N float arrays, the global NumPy RNG seeded exactly like
    cacc_env.py:22 and :169-176.  Stepping runs on the GPU kernel (E=1)."""

    def __init__(self, config, device='cuda'):
        self.batch = CACCBatchEnv(config, num_envs=1, device=device)
        b = self.batch
        self.agent, self.name, self.n_agent = b.agent, b.name, b.n_agent
        self.n_s_ls, self.n_a_ls, self.n_a = b.n_s_ls, b.n_a_ls, b.n_a
        self.neighbor_mask, self.distance_mask = b.neighbor_mask, b.distance_mask
        self.coop_gamma, self.T, self.dt, self.batch_size = b.coop_gamma, b.T, b.dt, b.batch_size
        self.seed = config.getint('seed')
        self.init_test_seeds([int(s) for s in config.get('test_seeds').split(',')])
        self.cur_episode = 0
        self.is_record = False
        self.a_map = [(0, 0), (0.5, 0), (0, 0.5), (0.5, 0.5)]
        self._nbr = [np.where(self.neighbor_mask[i] == 1)[0] for i in range(self.n_agent)]
        np.random.seed(self.seed)       # cacc_env.py:22 (model initialisation depends on it)

    train_mode = property(lambda self: self.batch.train_mode,
                          lambda self, f: setattr(self.batch, 'train_mode', f))

    # -- bookkeeping API of the reference (cacc_env.py:111-137, 244-251)
    def init_data(self, is_record, record_stats, output_path):
        self.is_record = is_record
        self.output_path = output_path
        if is_record:
            self.control_data, self.traffic_data = [], []

    def init_test_seeds(self, test_seeds):
        self.test_num = len(test_seeds)
        self.test_seeds = test_seeds

    def get_neighbor_action(self, action):
        action = np.asarray(action)
        return [action[self.neighbor_mask[i] == 1] for i in range(self.n_agent)]

    def get_fingerprint(self):
        return self.fp

    def update_fingerprint(self, fp):
        self.fp = fp

    def terminate(self):
        return

    def collect_tripinfo(self):
        return

    def _state_list(self):
        x = self.batch.obs[0].cpu().numpy().astype(np.float64)   # [8,15]
        fp = np.asarray(self.fp)
        out = []
        for i in range(self.n_agent):
            if self.agent.startswith('ia2c'):
                cur = [x[i, :N_FEAT * (1 + len(self._nbr[i]))]]
            else:
                cur = [x[i, :N_FEAT]]
            if self.agent == 'ia2c_fp':
                cur += [fp[j] for j in self._nbr[i]]
            out.append(np.concatenate(cur))
        return out

    def reset(self, gui=False, test_ind=-1):
        self.cur_episode += 1
        if self.train_mode:
            seed = self.seed
        elif test_ind < 0:
            seed = self.seed - 1
        else:
            seed = self.test_seeds[test_ind]
        np.random.seed(seed)
        self.seed += 1
        # the reference draws its single uniform only when seed != 0 (cacc_env.py:290-294, 311-314)
        U = np.random.rand() if self.seed else 0.5
        u0 = torch.tensor([U], dtype=torch.float32, device=self.batch.device)
        self.batch.reset(u0=u0)
        self.fp = np.ones((self.n_agent, self.n_a)) / self.n_a
        self._rewards = [0]
        self._hist = [self._phys()]
        return self._state_list()

    def _phys(self):
        b = self.batch
        s = torch.stack([b.h[0], b.v[0], b.u[0]]).cpu().numpy().astype(np.float64)
        return s

    def step(self, action):
        a = torch.as_tensor(np.asarray(action, dtype=np.uint8).reshape(1, -1), device=self.batch.device)
        _, reward, done, greward = self.batch.step(a)
        done = bool(done.item())
        global_reward = float(greward.item())
        reward = global_reward if self.coop_gamma < 0 else reward[0].cpu().numpy().astype(np.float64)
        self._rewards.append(global_reward)
        if self.is_record:
            self._hist.append(self._phys())
            self.control_data.append({'episode': self.cur_episode, 'time_sec': int(self.batch.t.item()) * self.dt,
                                      'step': int(self.batch.t.item()),
                                      'action': ','.join('%d' % x for x in action), 'reward': global_reward})
            if done:
                self._log_traffic_data()
        return self._state_list(), reward, done, global_reward

    def _log_traffic_data(self):
        """Per-episode traffic table with the column set of cacc_env.py:90-109."""
        hist = np.array(self._hist)
        hs, vs, us = hist[:, 0], hist[:, 1], hist[:, 2]
        df = pd.DataFrame()
        df['episode'] = np.ones(len(hs)) * self.cur_episode
        df['time_sec'] = np.arange(len(hs)) * self.dt
        df['reward'] = np.array(self._rewards)
        df['lead_headway_m'] = hs[:, 0]
        df['avg_headway_m'] = np.mean(hs[:, 1:], axis=1)
        df['std_headway_m'] = np.std(hs[:, 1:], axis=1)
        df['avg_speed_mps'] = np.mean(vs, axis=1)
        df['std_speed_mps'] = np.std(vs, axis=1)
        df['avg_accel_mps2'] = np.mean(us, axis=1)
        df['std_accel_mps2'] = np.std(us, axis=1)
        for i in range(self.n_agent):
            df['headway_%d_m' % (i + 1)] = hs[:, i]
            df['velocity_%d_mps' % (i + 1)] = vs[:, i]
            df['accel_%d_mps2' % (i + 1)] = us[:, i]
        self.traffic_data.append(df)

    def output_data(self):
        if not self.is_record:
            logging.error('Env: no record to output!')
            return
        pd.DataFrame(self.control_data).to_csv(self.output_path + ('%s_%s_control.csv' % (self.name, self.agent)))
        pd.concat(self.traffic_data).to_csv(self.output_path + ('%s_%s_traffic.csv' % (self.name, self.agent)))
