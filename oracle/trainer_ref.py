"""E = 1 restatement of the reference's rollout/training loop (utils.py:129-254) over the oracle
env (oracle/cacc_ref.py) and the oracle networks (oracle/nn_ref.py).  TEST INFRASTRUCTURE; it is
also what bench.py times as `cpu_baseline` (kind "port": TensorFlow 1.12 cannot run here, so the
reference-equivalent CPU path is this restatement -- never a TF measurement).
"""
import time

import numpy as np

from .cacc_ref import CaccBatchRef, CaccParams
from .nn_ref import REF_MODELS


class RefCaccEnv:
    """Reference duck-type (cacc_env.py:13-343) on top of CaccBatchRef(E=1, float64)."""

    def __init__(self, config):
        self.p = CaccParams(config=config)
        self.core = CaccBatchRef(self.p, E=1, dtype=np.float64)
        self.agent, self.name, self.n_agent = self.p.agent, self.p.name, self.p.n_agent
        self.T, self.coop_gamma, self.seed = self.p.T, self.p.coop_gamma, self.p.seed
        self.neighbor_mask, self.distance_mask = self.core.neighbor_mask, self.core.distance_mask
        self.n_a, self.n_a_ls = 4, [4] * self.n_agent
        self.n_s_ls = [5 if self.agent.startswith('ma2c') else 5 * (1 + int(self.neighbor_mask[i].sum()))
                       for i in range(self.n_agent)]
        self.train_mode = True
        np.random.seed(self.seed)                                   # cacc_env.py:22

    def _obs(self):
        return self.core.ref_obs_list(self.agent, fp=self.fp[None], e=0)

    def reset(self, test_ind=-1):
        seed = self.seed if self.train_mode else self.seed - 1      # cacc_env.py:169-176
        np.random.seed(seed)
        self.seed += 1
        self.core.train_mode = self.train_mode
        self.core.reset([np.random.rand()])
        self.fp = np.ones((self.n_agent, self.n_a)) / self.n_a
        return self._obs()

    def step(self, action):
        _, r, d, g = self.core.step(np.asarray(action)[None])
        reward = float(g[0]) if self.coop_gamma < 0 else r[0]
        return self._obs(), reward, bool(d[0]), float(g[0])

    def get_fingerprint(self):
        return self.fp

    def update_fingerprint(self, fp):
        self.fp = np.asarray(fp)

    def get_neighbor_action(self, action):
        return [np.asarray(action)[self.neighbor_mask[i] == 1] for i in range(self.n_agent)]


class RefTrainer:
    """explore / perform / run of utils.py:163-254, one replica, global np.random action draws."""

    def __init__(self, env, model):
        self.env, self.model, self.agent = env, model, env.agent
        self.n_step = model.n_step
        self.log = []           # (action, global_reward) per training step

    def _get_policy(self, ob, done, mode='train'):
        if self.agent.startswith('ma2c'):
            self.ps = self.env.get_fingerprint()
            policy = self.model.forward(ob, done, self.ps)
        else:
            policy = self.model.forward(ob, done)
        action = [np.random.choice(np.arange(len(pi)), p=pi) if mode == 'train' else np.argmax(pi) for pi in policy]
        return policy, np.array(action)

    def _get_value(self, ob, done, action):
        if self.agent.startswith('ma2c'):
            return self.model.forward(ob, done, self.ps, np.array(action), 'v')
        self.naction = self.env.get_neighbor_action(action)
        return self.model.forward(ob, done, self.naction, 'v')

    def explore(self, ob, done):
        for _ in range(self.n_step):
            policy, action = self._get_policy(ob, done)
            value = self._get_value(ob, done, action)
            self.env.update_fingerprint(policy)
            next_ob, reward, done, global_reward = self.env.step(action)
            self.log.append((action.copy(), global_reward))
            extra = self.ps if self.agent.startswith('ma2c') else self.naction
            self.model.add_transition(ob, extra, action, reward, value, done)
            if done:
                break
            ob = next_ob
        if done:
            R = np.zeros(self.env.n_agent)
        else:
            _, action = self._get_policy(ob, done)
            R = self._get_value(ob, done, action)
        return ob, done, R

    def run_batches(self, n_batches):
        """Runs `n_batches` rollout+update cycles (starting new episodes as needed, without the test
        episode); returns (env steps, seconds)."""
        t0 = time.perf_counter()
        steps, b = 0, 0
        done = True
        while b < n_batches:
            if done:
                ob = self.env.reset()
                self.model.reset()
            n0 = len(self.log)
            ob, done, R = self.explore(ob, done)
            self.model.backward(R, 0)
            steps += len(self.log) - n0
            b += 1
        return steps, time.perf_counter() - t0


def build(config_parser, dtype=None):
    import torch
    env = RefCaccEnv(config_parser['ENV_CONFIG'])
    model = REF_MODELS[env.agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma,
                                  config_parser['MODEL_CONFIG'], dtype=dtype or torch.float32)
    return env, model, RefTrainer(env, model)
