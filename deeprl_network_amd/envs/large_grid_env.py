"""Synthetic (SUMO-free) 5x5 ATSC grid on MI355X -- host mirror of the reference's
envs/large_grid_env.py + envs/atsc_env.py for the `atsc_large_grid` scenario.

The reference drives an external SUMO process over TraCI; this path keeps the reference's
*contract* (25 agents, 5 phases, 12-wide `wave` observation, queue reward, 5 s control / 2 s
yellow, 720-step episodes, the peak_flow demand schedule, neighbour / distance masks) and
replaces the microsimulation by the store-and-forward model specified in
oracle/grid_ref.py, stepped by csrc/grid.hip for E lock-stepped replicas.
"""
import ctypes

import numpy as np
import torch

from .. import _lib

N_NODE, N_FEAT, N_OBS, N_PHASE = 25, 12, 60, 5
OBJECTIVES = {'queue': 0, 'wait': 1, 'hybrid': 2}


def grid_masks():
    """neighbor_mask / distance_mask of large_grid_env.py:58-105 (node i = nt{i+1})."""
    idx = np.arange(N_NODE)
    r, c = idx // 5, idx % 5
    dist = (np.abs(r[:, None] - r[None, :]) + np.abs(c[:, None] - c[None, :])).astype(int)
    return (dist == 1).astype(int), dist


def grid_neighbor_order():
    """For every node its neighbours in the order the reference's `neighbor_map` lists them (large_grid_env.py:58-85: north,
    east, south, west, the absent ones skipped; node i = row * 5 + col = nt{i+1}, north = i + 5).  This -- not the ascending
    index -- is the order in which an IA2C / IA2C-FP agent's observation concatenates the neighbours' wave vectors and
    fingerprints (atsc_env.py:263-271); neighbour ACTIONS and the MA2C nets' gathers use the mask order (ascending)."""
    out = []
    for i in range(N_NODE):
        r, c = divmod(i, 5)
        cand = [(r + 1, c), (r, c + 1), (r - 1, c), (r, c - 1)]
        out.append([rr * 5 + cc for rr, cc in cand if 0 <= rr < 5 and 0 <= cc < 5])
    return out


# link -> physical lane (oracle/grid_ref.py LINK_LANE): the 12-wide wave vector counts duplicated lanes
_LANE_FIRST_LINK = (0, 3, 5, 6, 9, 11)


class LargeGridController:
    """The reference's rule-based `greedy` agent (large_grid_env.py:30-45): per node the phase whose served lanes hold the
    most vehicles.  The reference indexes a 6-lane observation [N, E lane 0, E lane 1, S, W lane 0, W lane 1]; this env's
    wave vector has one entry per signal LINK (12, duplicated lanes), so the six lanes are read at their first links.
    Same `forward(obs) -> actions` duck-type; `reset` / `load` exist so that Trainer.perform / main.py evaluate drive it."""
    name = 'greedy'
    n_step = 1

    def __init__(self, node_names=None):
        self.node_names = node_names

    @staticmethod
    def lane_counts(ob):
        ob = np.asarray(ob, dtype=np.float64).reshape(-1)
        return ob if len(ob) == 6 else ob[list(_LANE_FIRST_LINK)]

    def greedy(self, ob, node_name=None):
        q = self.lane_counts(ob)
        # phases of large_grid_env.py:25-26: N+S through, E+W left, E+W through, E approach, W approach
        return int(np.argmax([q[0] + q[3], q[2] + q[5], q[1] + q[4], q[1] + q[2], q[4] + q[5]]))

    def forward(self, obs):
        return [self.greedy(ob) for ob in obs]

    def reset(self):
        return

    def load(self, model_dir, checkpoint=None):
        return True


def grid_params_from_config(config):
    """ENV_CONFIG section -> nmarl_grid_params_t; keys of atsc_env.py:79-99 + large_grid_env.py:50-52."""
    if config.getint('control_interval_sec') != 5 or config.getint('yellow_interval_sec') != 2:
        raise _lib.NmarlError('the synthetic grid is specified for control 5 s / yellow 2 s')
    p = _lib.GridParams()
    obj = config.get('objective', fallback='queue')
    if obj not in OBJECTIVES:
        raise _lib.NmarlError('objective must be one of queue, wait, hybrid (atsc_env.py:87, 411-416), got %r' % obj)
    p.objective = OBJECTIVES[obj]
    p.coef_wait = config.getfloat('coef_wait', fallback=0.0)
    p.norm_wave = config.getfloat('norm_wave')
    p.clip_wave = config.getfloat('clip_wave')
    p.peak1 = config.getfloat('peak_flow1')
    p.peak2 = config.getfloat('peak_flow2')
    p.T = int(np.ceil(config.getint('episode_length_sec') / config.getint('control_interval_sec')))
    p.per_agent_reward = 0 if config.getfloat('coop_gamma') < 0 else 1
    return p


class LargeGridBatchEnv:
    def __init__(self, config, num_envs=1, device='cuda', env_id_base=0, seed=None):
        self.config = config
        self.E = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.NmarlError('LargeGridBatchEnv needs a HIP device; there is no CPU path')
        self.name = config.get('scenario')
        if self.name == 'large_grid':        # (config_greedy.ini's spelling): ONE name for the scenario -- Trainer / perform branch on 'atsc*'
            self.name = 'atsc_large_grid'
        self.agent = config.get('agent')
        self.coop_gamma = config.getfloat('coop_gamma')
        self.seed = config.getint('seed') if seed is None else int(seed)
        self.env_id_base = int(env_id_base)
        p = grid_params_from_config(config)
        self.params = p
        self.T = p.T
        self.n_agent = N_NODE
        self.n_a = N_PHASE
        self.n_a_ls = [N_PHASE] * N_NODE
        self.neighbor_mask, self.distance_mask = grid_masks()
        self.neighbor_order = grid_neighbor_order()
        self.n_s_ls = [N_FEAT * (1 + int(self.neighbor_mask[i].sum())) if self.agent.startswith('ia2c') else N_FEAT
                       for i in range(N_NODE)]
        self.train_mode = True
        E, d = self.E, self.device
        f32 = dict(dtype=torch.float32, device=d)
        self.q = torch.zeros(E, N_NODE, 6, **f32)
        self.transit = torch.zeros(E, N_NODE, 6, **f32)
        self.prev_action = torch.zeros(E, N_NODE, dtype=torch.uint8, device=d)
        self.t = torch.zeros(E, dtype=torch.int32, device=d)
        self.xi = torch.ones(E, 4, **f32)
        # `wait` / `hybrid` objectives: the front vehicle's standing time per lane (oracle/grid_ref.py step 6)
        self.head_wait = torch.zeros(E, N_NODE, 6, **f32) if p.objective else None
        if self.head_wait is not None:
            p.head_wait = self.head_wait.data_ptr()
        self.obs = torch.zeros(E, N_NODE, N_OBS, **f32)
        self.reward = torch.zeros((E, N_NODE) if p.per_agent_reward else (E,), **f32)
        self.done = torch.zeros(E, dtype=torch.uint8, device=d)
        self.global_reward = torch.zeros(E, **f32)
        self.episode = torch.zeros(E, dtype=torch.int32, device=d)
        self.batch_size = None     # episodes end at T only; any n_step dividing T works

    def state_tensors(self):
        return [self.q, self.transit, self.prev_action, self.t, self.xi, self.obs, self.episode, self.done] + \
            ([self.head_wait] if self.head_wait is not None else [])

    compact_obs = False

    def set_compact_obs(self, flag=True):
        """Compact observation [E,25,12]: every node's OWN wave vector -- what the reference hands an MA2C agent
        (atsc_env.py:253-262) -- instead of the gathered [E,25,60] slab; the consumer gathers the neighbours
        (agents/policies.py `_ob_part`).  Batched engine only: the E = 1 reference duck-type keeps the slab."""
        self.compact_obs = bool(flag)
        self.params.compact_obs = 1 if flag else 0
        self.obs = torch.zeros(self.E, N_NODE, N_FEAT if flag else N_OBS, dtype=torch.float32, device=self.device)
        return True

    def reset(self, mask=None, u0=None):
        P = _lib.ptr
        rc = _lib.lib.nmarl_grid_reset(ctypes.byref(self.params), self.E, P(mask, torch.uint8), P(u0, torch.float32),
                                       self.seed, self.env_id_base, P(self.episode), P(self.q), P(self.transit),
                                       P(self.prev_action), P(self.t), P(self.xi), P(self.obs), _lib.stream())
        _lib.check(rc, 'nmarl_grid_reset')
        return self.obs

    _words = None

    def inkernel_step_supported(self):
        """CommNet's one-launch lock-step can run this env's step as a role of the same launch (csrc/lstm_mfma.hip GENV): compact
        observation, queue objective, and compute units left idle by the LSTM blocks."""
        from .. import ops
        return self.compact_obs and not self.params.objective and ops.step_grid_env_blocks(N_NODE, self.E) > 0

    def inkernel_step(self, auto_reset=False, obs_out=None, reward_out=None, done_out=None, greward_out=None):
        """Arguments of `step` for the policy's lock-step launch to run the env step itself, on the actions it draws
        (ops._step_x msg['genv'], nmarl_lstm_step_x_msg_grid): same state tensors, same outputs."""
        if not self.compact_obs or self.params.objective:
            raise _lib.NmarlError('the in-launch grid env step writes the compact observation and knows the queue objective')
        if self._words is None:          # hand-off words of the launch ([E][2] u64): zeroed once, every launch leaves them zero
            self._words = torch.zeros(_lib.lib.nmarl_lstm_step_grid_words(self.E), dtype=torch.int64, device=self.device)
        return dict(params=self.params, q=self.q, transit=self.transit, prev_action=self.prev_action, t=self.t, xi=self.xi,
                    obs_out=self.obs if obs_out is None else obs_out, reward=self.reward if reward_out is None else reward_out,
                    done=self.done if done_out is None else done_out,
                    global_reward=self.global_reward if greward_out is None else greward_out, auto_reset=bool(auto_reset),
                    seed=self.seed, env_id_base=self.env_id_base, episode=self.episode, words=self._words)

    def clear_inkernel_words(self):
        """After a hand-off time-out (the launch gave up waiting and may have left arrivals in the words)."""
        if self._words is not None:
            self._words.zero_()

    def step(self, action, auto_reset=False, obs_out=None, reward_out=None, done_out=None, greward_out=None):
        P = _lib.ptr
        obs = self.obs if obs_out is None else obs_out
        reward = self.reward if reward_out is None else reward_out
        done = self.done if done_out is None else done_out
        greward = self.global_reward if greward_out is None else greward_out
        rc = _lib.lib.nmarl_grid_step(ctypes.byref(self.params), self.E, P(action, torch.uint8), P(self.q),
                                      P(self.transit), P(self.prev_action), P(self.t), P(self.xi),
                                      P(obs, torch.float32), P(reward, torch.float32), P(done, torch.uint8),
                                      P(greward, torch.float32), 1 if auto_reset else 0, self.seed,
                                      self.env_id_base, P(self.episode), _lib.stream())
        _lib.check(rc, 'nmarl_grid_step')
        return obs, reward, done, greward


class LargeGridEnv:
    """Reference duck-type (atsc_env.py:77-524 / large_grid_env.py:48-137) for ONE replica.
    Observation lists: `ma2c*` / `greedy` 12 wave features; `ia2c*` own + neighbours' in the reference's `neighbor_map` list
    order (north, east, south, west: atsc_env.py:263-269, `neighbor_order`) (+ the neighbours' fingerprints in the same order
    for ia2c_fp); neighbour actions in mask order (ascending, atsc_env.py:132-136)."""

    def __init__(self, config, port=0, device='cuda', **_):
        self.batch = LargeGridBatchEnv(config, num_envs=1, device=device)
        b = self.batch
        self.name, self.agent, self.coop_gamma, self.T = b.name, b.agent, b.coop_gamma, b.T
        self.n_agent, self.n_a, self.n_a_ls, self.n_s_ls = b.n_agent, b.n_a, b.n_a_ls, b.n_s_ls
        self.neighbor_mask, self.distance_mask = b.neighbor_mask, b.distance_mask
        self.neighbor_order = b.neighbor_order
        self.node_names = ['nt%d' % (i + 1) for i in range(self.n_agent)]
        self.seed = config.getint('seed')
        self.control_interval_sec = config.getint('control_interval_sec')
        self.init_test_seeds([int(s) for s in config.get('test_seeds').split(',')])
        self.cur_episode = 0
        self.train_mode = True
        self.is_record = False
        self._nbr = [np.where(self.neighbor_mask[i] == 1)[0] for i in range(self.n_agent)]

    def init_data(self, is_record, record_stats, output_path):
        self.is_record, self.output_path = is_record, output_path
        if is_record:
            self.control_data = []

    def init_test_seeds(self, test_seeds):
        self.test_num, self.test_seeds = len(test_seeds), test_seeds

    def get_neighbor_action(self, action):
        action = np.asarray(action)
        return [action[self.neighbor_mask[i] == 1] for i in range(self.n_agent)]

    def get_fingerprint(self):
        return self.fp

    def update_fingerprint(self, policy):
        self.fp = policy

    def terminate(self):
        return

    def collect_tripinfo(self):
        return

    def output_data(self):
        if self.is_record:
            import pandas as pd
            pd.DataFrame(self.control_data).to_csv(self.output_path + ('%s_%s_control.csv' % (self.name, self.agent)))

    def _state_list(self):
        own = self.batch.obs[0, :, :N_FEAT].cpu().numpy().astype(np.float64)      # every node's own wave vector leads its row
        out = []
        for i in range(self.n_agent):
            cur = [own[i]]
            if self.agent.startswith('ia2c'):
                cur += [own[j] for j in self.neighbor_order[i]]
            if self.agent == 'ia2c_fp':
                cur += [np.asarray(self.fp[j]) for j in self.neighbor_order[i]]
            out.append(np.concatenate(cur))
        return out

    def reset(self, gui=False, test_ind=0):
        seed = self.seed if self.train_mode else self.test_seeds[test_ind]      # atsc_env.py:167-170
        self.batch.seed = seed
        self.batch.episode.zero_()
        self.batch.reset()
        self.cur_episode += 1
        self.fp = [np.ones(self.n_a) / self.n_a for _ in range(self.n_agent)]
        self.seed += 1
        return self._state_list()

    def step(self, action):
        a = torch.as_tensor(np.asarray(action, dtype=np.uint8).reshape(1, -1), device=self.batch.device)
        _, reward, done, g = self.batch.step(a)
        global_reward = float(g.item())
        done = bool(done.item())
        if self.agent == 'greedy' or self.coop_gamma < 0:                           # atsc_env.py:205-206
            reward = global_reward
        else:
            reward = reward[0].cpu().numpy().astype(np.float64)
        if self.is_record:
            sec = int(self.batch.t.item()) * self.control_interval_sec
            self.control_data.append({'episode': self.cur_episode, 'time_sec': sec,
                                      'step': sec / self.control_interval_sec,
                                      'action': ','.join('%d' % x for x in action), 'reward': global_reward})
        return self._state_list(), reward, done, global_reward
