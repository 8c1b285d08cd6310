"""Synthetic (SUMO-free) Monaco-like ATSC network on MI355X -- host mirror of the reference's
envs/real_net_env.py + envs/atsc_env.py for the `atsc_real_net` scenario: 28 HETEROGENEOUS agents (2..6 phases over
2..22 signal links, 0..4 listed neighbours).

The reference drives an external SUMO process on a net file that is not in its repository; this path keeps the
reference's *contract* -- node set, directed neighbour lists, phase sets (real_net_env.py:21-69), node order and
neighbour / BFS distance masks (152-195), 5 s control / 2 s yellow, 720-step episodes, `wave` observation, queue
reward with per-agent spatial discount, the flow_rate demand schedule (real_net_data/build_file.py:70-96) -- and
replaces the microsimulation by the store-and-forward link-graph model specified in oracle/realnet_ref.py, stepped
by csrc/realnet.hip for E lock-stepped replicas.  Observations come out already padded to the widest node and
gathered over the listed neighbours: the input layout of the heterogeneous (identical=False) nets.
"""
import ctypes

import numpy as np
import torch

from .. import _lib

# (node, phase-set key, listed neighbours): real_net_env.py:21-49
NODE_DEFS = (
    ('10026', '6.0', ('9431', '9561', 'cluster_9563_9597', '9531')),
    ('8794', '4.0', ('cluster_8985_9609', '9837', '9058', 'cluster_9563_9597')),
    ('8940', '2.1', ('9007', '9429')),
    ('8996', '2.2', ()),
    ('9007', '2.3', ('9309', '8940')),
    ('9058', '4.0', ('cluster_8985_9609', '8794', 'joinedS_0')),
    ('9153', '2.0', ('9643',)),
    ('9309', '4.0', ('9466', '9007', 'cluster_9043_9052')),
    ('9413', '2.3', ('9721', '9837')),
    ('9429', '5.0', ('cluster_9043_9052', '8940')),
    ('9431', '2.4', ('9721', '9884', '9561', '10026')),
    ('9433', '2.5', ()),
    ('9466', '4.0', ('9309', 'joinedS_0')),
    ('9480', '2.3', ()),
    ('9531', '2.6', ('joinedS_1',)),
    ('9561', '4.0', ('cluster_9389_9689', '10026')),
    ('9643', '2.3', ('9153',)),
    ('9713', '3.0', ('9721',)),
    ('9721', '6.0', ('9431', '9713', '9413')),
    ('9837', '3.1', ('9413', '8794', 'cluster_8985_9609')),
    ('9884', '2.7', ('9713', 'cluster_9389_9689')),
    ('cluster_8751_9630', '4.0', ()),
    ('cluster_8985_9609', '4.0', ('9837', '8794', '9058')),
    ('cluster_9043_9052', '4.1', ('cluster_9563_9597', '10026', 'joinedS_1')),
    ('cluster_9389_9689', '4.0', ('cluster_8751_9630', '9884', '9561', '8996')),
    ('cluster_9563_9597', '4.2', ('10026', '8794', 'joinedS_0', 'cluster_9043_9052')),
    ('joinedS_0', '6.1', ('9058', 'cluster_9563_9597', '9466')),
    ('joinedS_1', '3.2', ('9531', '9429')),
)
# phase sets over the node's signal links: real_net_env.py:51-69
PHASE_SETS = {
    '2.0': ('GGrrr', 'ggGGG'), '2.1': ('GGGrrr', 'rrGGGg'), '2.2': ('Grr', 'gGG'), '2.3': ('GGGgrr', 'GrrrGG'),
    '2.4': ('GGGGrr', 'rrrrGG'), '2.5': ('Gg', 'rG'), '2.6': ('GGGg', 'rrrG'), '2.7': ('GGg', 'rrG'),
    '3.0': ('GGgrrrGGg', 'rrGrrrrrG', 'rrrGGGGrr'), '3.1': ('GgrrGG', 'rGrrrr', 'rrGGGr'),
    '3.2': ('GGGGrrrGG', 'rrrrGGGGr', 'GGGGrrGGr'),
    '4.0': ('GGgrrrGGgrrr', 'rrrGGgrrrGGg', 'rrGrrrrrGrrr', 'rrrrrGrrrrrG'),
    '4.1': ('GGgrrGGGrrr', 'rrGrrrrrrrr', 'rrrGgrrrGGg', 'rrrrGrrrrrG'),
    '4.2': ('GGGGrrrrrrrr', 'GGggrrGGggrr', 'rrrGGGGrrrrr', 'grrGGggrrGGg'),
    '5.0': ('GGGGgrrrrGGGggrrrr', 'grrrGrrrrgrrGGrrrr', 'GGGGGrrrrrrrrrrrrr', 'rrrrrrrrrGGGGGrrrr', 'rrrrrGGggrrrrrggGg'),
    '6.0': ('GGGgrrrGGGgrrr', 'rrrGrrrrrrGrrr', 'GGGGrrrrrrrrrr', 'rrrrrrrrrrGGGG', 'rrrrGGgrrrrGGg', 'rrrrrrGrrrrrrG'),
    '6.1': ('GGgrrGGGrrrGGGgrrrGGGg', 'rrGrrrrrrrrrrrGrrrrrrG', 'GGGrrrrrGGgrrrrGGgrrrr', 'GGGrrrrrrrGrrrrrrGrrrr',
            'rrrGGGrrrrrrrrrrrrGGGG', 'rrrGGGrrrrrGGGgrrrGGGg'),
}
N_GROUP = 4


class NetTopology:
    """Static arrays of the network (host NumPy + device copies + the nmarl_net_topo_t handed to the kernels)."""

    def __init__(self, device):
        defs = {name: (key, nbrs) for name, key, nbrs in NODE_DEFS}
        self.node_names = sorted(defs)                                   # real_net_env.py:189
        N = self.N = len(self.node_names)
        pos = {name: i for i, name in enumerate(self.node_names)}
        self.phases = [PHASE_SETS[defs[name][0]] for name in self.node_names]
        self.n_a_ls = [len(p) for p in self.phases]
        self.n_s_ls = [len(p[0]) for p in self.phases]                   # one `wave` entry per signal link
        self.A, self.L = max(self.n_a_ls), max(self.n_s_ls)
        listed = [[pos[m] for m in defs[name][1]] for name in self.node_names]
        self.neighbor_mask = np.zeros((N, N), dtype=int)                 # real_net_env.py:175-181 (directed)
        for i, js in enumerate(listed):
            self.neighbor_mask[i, js] = 1
        self.distance_mask = -np.ones((N, N), dtype=int)                 # BFS, -1 = unreachable (152-187)
        for i in range(N):
            self.distance_mask[i, i] = 0
            frontier, d = [i], 0
            while frontier:
                d += 1
                nxt = []
                for u in frontier:
                    for v in listed[u]:
                        if self.distance_mask[i, v] < 0:
                            self.distance_mask[i, v] = d
                            nxt.append(v)
                frontier = nxt
        self.nbrs = [sorted(js) for js in listed]
        self.m_max = max(len(js) for js in self.nbrs)
        code = {'r': 0, 'G': 1, 'g': 2}
        green = np.zeros((N, self.A, self.L), dtype=np.uint8)
        for i, p in enumerate(self.phases):
            for a, s in enumerate(p):
                green[i, a, :len(s)] = [code[ch] for ch in s]
        src = -np.ones((N, self.L), dtype=np.int32)
        for i in range(N):
            m = len(self.nbrs[i])
            for k in range(self.n_s_ls[i]):
                if k % (m + 1) < m:
                    src[i, k] = self.nbrs[i][k % (m + 1)]
        fan = np.array([(src == j).sum() for j in range(N)], dtype=np.int32)
        group = -np.ones((N, self.L), dtype=np.int32)
        for i in range(N):
            for k in range(self.n_s_ls[i]):
                if src[i, k] < 0:
                    group[i, k] = (i + k) % N_GROUP
        n_ext = np.array([(group == g).sum() for g in range(N_GROUP)])
        ext_share = np.where(group >= 0, 1.0 / n_ext[np.maximum(group, 0)], 0.0).astype(np.float32)
        dn_ptr, dn_pair = [0], []
        for j in range(N):
            for i in range(N):
                for k in range(self.n_s_ls[i]):
                    if src[i, k] == j:
                        dn_pair.append((i << 8) | k)
            dn_ptr.append(len(dn_pair))
        nbr_idx = -np.ones((N, self.m_max), dtype=np.int32)
        for i, js in enumerate(self.nbrs):
            nbr_idx[i, :len(js)] = js
        self.host = dict(n_s=np.array(self.n_s_ls, dtype=np.int32), green=green, src=src, fan=fan, group=group,
                         ext_share=ext_share, dn_ptr=np.array(dn_ptr, dtype=np.int32),
                         dn_pair=np.array(dn_pair if dn_pair else [0], dtype=np.int32), nbr_idx=nbr_idx)
        # the packed image the step kernel copies into LDS (include/nmarl.h NMARL_NET_OFF_*: rows padded to 24 links)
        NM, LM, AM = 32, 24, 8
        if N > NM or self.L > LM or self.A > AM or self.m_max > 8:
            raise _lib.NmarlError('network exceeds the kernel limits (32 nodes, 24 links, 8 phases, 8 neighbours)')
        img = np.zeros(_lib.NET_IMAGE_BYTES, dtype=np.uint8)
        off = _lib.NET_OFF

        def put(name, arr):
            b = np.ascontiguousarray(arr).view(np.uint8).ravel()
            img[off[name]:off[name] + b.size] = b
        g = np.zeros((NM, AM, LM), dtype=np.uint8)
        g[:N, :self.A, :self.L] = green
        put('green', g)
        a16 = -np.ones((NM, LM), dtype=np.int16)
        a16[:N, :self.L] = src
        put('src', a16)
        a8 = -np.ones((NM, LM), dtype=np.int8)
        a8[:N, :self.L] = group
        put('group', a8)
        f = np.zeros((NM, LM), dtype=np.float32)
        f[:N, :self.L] = ext_share
        put('share', f)
        ff = np.zeros(NM, dtype=np.float32)
        ff[:N] = fan
        put('fan', ff)
        put('dnptr', np.array(dn_ptr + [dn_ptr[-1]] * (NM + 1 - len(dn_ptr)), dtype=np.int16))
        pairs = np.zeros(NM * LM, dtype=np.int16)
        pairs[:len(dn_pair)] = [(v >> 8) * LM + (v & 255) for v in dn_pair]
        put('dnpair', pairs)
        nb8 = -np.ones((NM, 8), dtype=np.int8)
        nb8[:N, :self.m_max] = nbr_idx
        put('nbr', nb8)
        self.host['image'] = img
        self.dev = {'n_s': torch.from_numpy(self.host['n_s']).to(device), 'image': torch.from_numpy(img).to(device),
                    'nbr_idx': torch.from_numpy(nbr_idx).to(device)}
        t = self.c = _lib.NetTopo()
        t.N, t.L, t.A, t.m_max = N, self.L, self.A, self.m_max
        t.n_s, t.image = self.dev['n_s'].data_ptr(), self.dev['image'].data_ptr()


def net_params_from_config(config):
    """ENV_CONFIG section -> nmarl_net_params_t; keys of atsc_env.py:79-99 + real_net_env.py:147."""
    if config.getint('control_interval_sec') != 5 or config.getint('yellow_interval_sec') != 2:
        raise _lib.NmarlError('the synthetic network is specified for control 5 s / yellow 2 s')
    if config.get('objective') != 'queue':
        raise NotImplementedError('only the `queue` objective of the shipped real-net configs is modelled')
    p = _lib.NetParams()
    p.norm_wave = config.getfloat('norm_wave')
    p.clip_wave = config.getfloat('clip_wave')
    p.flow_rate = config.getfloat('flow_rate')
    p.T = int(np.ceil(config.getint('episode_length_sec') / config.getint('control_interval_sec')))
    p.per_agent_reward = 0 if config.getfloat('coop_gamma') < 0 else 1
    return p


class RealNetBatchEnv:
    def __init__(self, config, num_envs=1, device='cuda', env_id_base=0, seed=None):
        self.config = config
        self.E = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.NmarlError('RealNetBatchEnv needs a HIP device; there is no CPU path')
        self.name = config.get('scenario')
        self.agent = config.get('agent')
        self.coop_gamma = config.getfloat('coop_gamma')
        self.seed = config.getint('seed') if seed is None else int(seed)
        self.env_id_base = int(env_id_base)
        self.params = net_params_from_config(config)
        self.T = self.params.T
        tp = self.topo = NetTopology(self.device)
        self.n_agent, self.n_a, self.n_a_ls = tp.N, tp.A, list(tp.n_a_ls)
        self.n_feat, self.n_feat_ls = tp.L, list(tp.n_s_ls)               # own observation widths (heterogeneous)
        self.neighbor_mask, self.distance_mask = tp.neighbor_mask, tp.distance_mask
        if self.agent.startswith('ma2c'):
            self.n_s_ls = list(tp.n_s_ls)
        else:                                                            # own + listed neighbours' (atsc_env.py:263-269)
            self.n_s_ls = [tp.n_s_ls[i] + sum(tp.n_s_ls[j] for j in tp.nbrs[i]) for i in range(tp.N)]
        self.train_mode = True
        E, d, N, L = self.E, self.device, tp.N, tp.L
        f32 = dict(dtype=torch.float32, device=d)
        self.q = torch.zeros(E, N, L, **f32)
        self.transit = torch.zeros(E, N, L, **f32)
        self.prev_action = torch.zeros(E, N, dtype=torch.uint8, device=d)
        self.t = torch.zeros(E, dtype=torch.int32, device=d)
        self.xi = torch.ones(E, N_GROUP, **f32)
        self.obs = torch.zeros(E, N, L * (1 + tp.m_max), **f32)
        self.reward = torch.zeros((E, N) if self.params.per_agent_reward else (E,), **f32)
        self.done = torch.zeros(E, dtype=torch.uint8, device=d)
        self.global_reward = torch.zeros(E, **f32)
        self.episode = torch.zeros(E, dtype=torch.int32, device=d)
        self.batch_size = None     # episodes end at T only; any n_step dividing T works

    def state_tensors(self):
        return [self.q, self.transit, self.prev_action, self.t, self.xi, self.obs, self.episode, self.done]

    def reset(self, mask=None, u0=None):
        P = _lib.ptr
        rc = _lib.lib.nmarl_net_reset(ctypes.byref(self.topo.c), self.E, P(mask, torch.uint8), P(u0, torch.float32),
                                      self.seed, self.env_id_base, P(self.episode), P(self.q), P(self.transit),
                                      P(self.prev_action), P(self.t), P(self.xi), P(self.obs), _lib.stream())
        _lib.check(rc, 'nmarl_net_reset')
        return self.obs

    def step(self, action, auto_reset=False, obs_out=None, reward_out=None, done_out=None, greward_out=None):
        P = _lib.ptr
        obs = self.obs if obs_out is None else obs_out
        reward = self.reward if reward_out is None else reward_out
        done = self.done if done_out is None else done_out
        greward = self.global_reward if greward_out is None else greward_out
        rc = _lib.lib.nmarl_net_step(ctypes.byref(self.params), ctypes.byref(self.topo.c), self.E, P(action, torch.uint8),
                                     P(self.q), P(self.transit), P(self.prev_action), P(self.t), P(self.xi),
                                     P(obs, torch.float32), P(reward, torch.float32), P(done, torch.uint8),
                                     P(greward, torch.float32), 1 if auto_reset else 0, self.seed, self.env_id_base,
                                     P(self.episode), _lib.stream())
        _lib.check(rc, 'nmarl_net_step')
        return obs, reward, done, greward


class RealNetEnv:
    """Reference duck-type (atsc_env.py:77-524 / real_net_env.py:145-198) for ONE replica: ragged observation lists
    (`ma2c*`: the node's own links; `ia2c*`: own + listed neighbours' in ascending node index, + their fingerprints
    for ia2c_fp), ragged fingerprints, per-agent rewards."""

    def __init__(self, config, port=0, device='cuda', **_):
        self.batch = RealNetBatchEnv(config, num_envs=1, device=device)
        b = self.batch
        self.name, self.agent, self.coop_gamma, self.T = b.name, b.agent, b.coop_gamma, b.T
        self.n_agent, self.n_a, self.n_a_ls, self.n_s_ls = b.n_agent, b.n_a, b.n_a_ls, b.n_s_ls
        self.n_feat_ls = b.n_feat_ls
        self.node_names = b.topo.node_names
        self.neighbor_mask, self.distance_mask = b.neighbor_mask, b.distance_mask
        self.seed = config.getint('seed')
        self.control_interval_sec = config.getint('control_interval_sec')
        self.init_test_seeds([int(s) for s in config.get('test_seeds').split(',')])
        self.cur_episode = 0
        self.train_mode = True
        self.is_record = False
        self._nbr = b.topo.nbrs

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
        L = self.batch.n_feat
        x = self.batch.obs[0].cpu().numpy().astype(np.float64)          # [N, L*(1+m_max)], slots L wide
        out = []
        for i in range(self.n_agent):
            cur = [x[i, :self.n_feat_ls[i]]]
            if self.agent.startswith('ia2c'):
                cur += [x[i, (k + 1) * L:(k + 1) * L + self.n_feat_ls[j]] for k, j in enumerate(self._nbr[i])]
            if self.agent == 'ia2c_fp':
                cur += [np.asarray(self.fp[j]) for j in self._nbr[i]]
            out.append(np.concatenate(cur))
        return out

    def reset(self, gui=False, test_ind=0):
        seed = self.seed if self.train_mode else self.test_seeds[test_ind]      # atsc_env.py:167-170
        self.batch.seed = seed
        self.batch.episode.zero_()
        self.batch.reset()
        self.cur_episode += 1
        self.fp = [np.ones(a) / a for a in self.n_a_ls]                  # atsc_env.py:498-499
        self.seed += 1
        return self._state_list()

    def step(self, action):
        a = torch.as_tensor(np.asarray(action, dtype=np.uint8).reshape(1, -1), device=self.batch.device)
        _, reward, done, g = self.batch.step(a)
        global_reward = float(g.item())
        done = bool(done.item())
        if self.coop_gamma < 0 and self.train_mode:
            reward = global_reward
        else:
            reward = reward[0].cpu().numpy().astype(np.float64) if self.coop_gamma >= 0 else global_reward
        if self.is_record:
            sec = int(self.batch.t.item()) * self.control_interval_sec
            self.control_data.append({'episode': self.cur_episode, 'time_sec': sec,
                                      'step': sec / self.control_interval_sec,
                                      'action': ','.join('%d' % x for x in action), 'reward': global_reward})
        return self._state_list(), reward, done, global_reward
