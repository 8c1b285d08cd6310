"""End-to-end golden: the REAL reference Trainer (utils.py:100-254) + REAL CACCEnv + REAL model code
(on the fake-TF shim, float64) for one training episode and its deterministic test episode, E = 1,
global NumPy RNG -- exactly what `python main.py train` does for the first episode; and (`run_episodes`) for the first
THREE training episodes with their test episodes, which pins the episode seam: the second `env.reset()` (seed0 + 2,
cacc_env.py:166-189), `model.reset()`, the carry of states_bw / the RMSProp slots across episodes and across the
test episodes in between (utils.py:213-254, agents/policies.py:115, 151-154).

    python tests/golden/make_golden_e2e.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'tf1_shim'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import tensorflow as tf  # noqa: E402  (shim)
from envs.cacc_env import CACCEnv  # noqa: E402  (reference)
from agents.models import IA2C, IA2C_FP, MA2C_NC, MA2C_IC3  # noqa: E402
from utils import Counter, Trainer  # noqa: E402  (reference root utils.py)
from helpers import cacc_config  # noqa: E402
from make_golden_nn import var_stats  # noqa: E402

CLS = {'ia2c': IA2C, 'ia2c_fp': IA2C_FP, 'ma2c_nc': MA2C_NC, 'ma2c_ic3': MA2C_IC3}


def run(name, agent, scenario, seed, reward_norm):
    cp = cacc_config(agent=agent, scenario=scenario, seed=seed, n_step=60, reward_norm=reward_norm, total_step=60)
    env = CACCEnv(cp['ENV_CONFIG'])                      # seeds np.random (cacc_env.py:22)
    counter = Counter(60, 10 ** 9, 10 ** 9)              # total_step 60 -> exactly one episode (+ its test episode)
    model = CLS[agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 60,
                       cp['MODEL_CONFIG'], seed=seed)
    log = {'a': [], 'g': [], 'train': []}
    orig_step = env.step

    def step(action):
        out = orig_step(action)
        log['a'].append(np.array(action).copy())
        log['g'].append(out[3])
        log['train'].append(env.train_mode)
        return out
    env.step = step
    trainer = Trainer(env, model, counter, tf.summary.FileWriter(None), output_path=None)
    trainer.output_path = '/tmp/e2e_'
    trainer.run()
    tr = np.array(log['train'])
    acts, g = np.array(log['a']), np.array(log['g'])
    out = dict(train_actions=acts[tr], train_rewards=g[tr], test_actions=acts[~tr], test_rewards=g[~tr],
               logged_mean=trainer.data[0]['avg_reward'], logged_std=trainer.data[0]['std_reward'],
               logged_step=trainer.data[0]['step'], stats=var_stats(tf.global_variables()),
               agent=agent, scenario=scenario, seed=seed, reward_norm=reward_norm)
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % name), **out)
    print('%-18s train steps %d (sum g %.3f) test steps %d mean %.4f' % (
        name, tr.sum(), g[tr].sum(), (~tr).sum(), out['logged_mean']))


def run_episodes(name, agent, scenario, seed, reward_norm, episodes=2):
    """The first `episodes` training episodes + their test episodes.  The reference loop stops when the counter says so
    (utils.py:214): the counter's `stop` flag (utils.py:94-97) is raised when the last wanted training episode begins.
    Runs the shim single-threaded: the float64 reductions of torch-CPU depend on the thread count in their last bits, and
    with the host's default (8) a busy machine makes the tiny per-agent ops spin for minutes."""
    import torch
    torch.set_num_threads(1)
    cp = cacc_config(agent=agent, scenario=scenario, seed=seed, n_step=60, reward_norm=reward_norm, total_step=10 ** 9)
    env = CACCEnv(cp['ENV_CONFIG'])
    counter = Counter(10 ** 9, 10 ** 9, 10 ** 9)
    model = CLS[agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                       cp['MODEL_CONFIG'], seed=seed)
    log = {'a': [], 'g': [], 'train': [], 'ep': [], 'h0': []}
    orig_step, orig_reset = env.step, env.reset
    n_train = [0]

    def reset(*a, **k):
        if env.train_mode:
            n_train[0] += 1
            if n_train[0] == episodes:
                counter.stop = True
        ob = orig_reset(*a, **k)
        log['h0'].append(np.concatenate([env.hs[0], env.vs[0]]))       # initial headways / speeds of every episode
        return ob

    def step(action):
        out = orig_step(action)
        log['a'].append(np.array(action).copy())
        log['g'].append(out[3])
        log['train'].append(env.train_mode)
        log['ep'].append(n_train[0])
        return out
    env.step, env.reset = step, reset
    trainer = Trainer(env, model, counter, tf.summary.FileWriter(None), output_path=None)
    trainer.output_path = '/tmp/e2e_multi_'
    trainer.run()
    assert len(trainer.data) == episodes
    out = dict(actions=np.array(log['a']), rewards=np.array(log['g']), train=np.array(log['train']),
               episode=np.array(log['ep']), init_state=np.array(log['h0']),
               logged=np.array([[d['avg_reward'], d['std_reward'], d['step']] for d in trainer.data]),
               stats=var_stats(tf.global_variables()), agent=agent, scenario=scenario, seed=seed, reward_norm=reward_norm)
    np.savez_compressed(os.path.join(OUT, 'e2e_multi_%s.npz' % name), **out)
    print('%-18s %d episodes: steps %d (train %d), logged %s' % (name, episodes, len(log['a']), int(out['train'].sum()),
                                                                   out['logged'][:, 0]))


if __name__ == '__main__':
    if '--out' in sys.argv:                       # regeneration check (tests/test_golden_regen.py): write elsewhere
        OUT = sys.argv[sys.argv.index('--out') + 1]
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else 'single,multi'
    if 'single' in only:
        run('ia2c_fp_catchup', 'ia2c_fp', 'catchup', 12, 800.0)
        run('ma2c_nc_slowdown', 'ma2c_nc', 'slowdown', 12, 5000.0)
    if 'multi' in only:
        run_episodes('ma2c_nc_slowdown', 'ma2c_nc', 'slowdown', 12, 5000.0, episodes=3)
