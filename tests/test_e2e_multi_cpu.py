"""The episode seam of the reference loop over THREE training episodes and their test episodes
(tests/golden/e2e_multi_ma2c_nc_slowdown.npz: the REAL reference Trainer + CACCEnv + MA2C_NC on the fake-TF shim):
second / third `env.reset()` (seeds 12, 14, 16; the test episodes reuse the training seed, cacc_env.py:166-176),
`model.reset()`, states_bw and the RMSProp slots carried across episodes (utils.py:213-254).

  * the CPU port used for the E = 1 learning runs (oracle/trainer_ref.py + tests/learning/port_full_schedule.py) replays it
    action for action;
  * the PRODUCT's reference-API path (agents.models.MA2C_NC forward / add_transition / backward / reset + utils.Trainer)
    replays it on the CPU emulation of the HIP ops, driven by the oracle env in the reference's duck-type.
The same fixture through the HIP kernels and the product env: tests/test_gpu_e2e.py."""
import os
import sys

import numpy as np
import torch

from helpers import GOLDEN, cacc_config, load_npz, var_stats_from_named

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'learning'))


def _golden():
    z = load_npz(os.path.join(GOLDEN, 'e2e_multi_ma2c_nc_slowdown.npz'))
    cp = cacc_config(agent=str(z['agent']), scenario=str(z['scenario']), seed=int(z['seed']), n_step=60,
                     reward_norm=float(z['reward_norm']), total_step=10 ** 9)
    return z, cp


def test_fixture_covers_three_episodes_and_their_seeds():
    z, _ = _golden()
    assert z['logged'].shape == (3, 3) and list(z['logged'][:, 2]) == [300, 600, 900]
    ep, tr = z['episode'], z['train']
    assert set(ep) == {1, 2, 3} and all(tr[ep == k].any() and (~tr[ep == k]).any() for k in (1, 2, 3))
    # reset order: train 1, test 1, train 2, ...; a test episode starts from its training episode's initial state
    s = z['init_state']
    assert s.shape == (6, 16)
    for k in range(3):
        np.testing.assert_array_equal(s[2 * k], s[2 * k + 1])
    assert not np.array_equal(s[0], s[2]) and not np.array_equal(s[2], s[4])


def test_cpu_port_replays_three_reference_episodes():
    import port_full_schedule as pfs
    from oracle import trainer_ref
    z, cp = _golden()
    torch.set_num_threads(1)
    env, model, tr = trainer_ref.build(cp, dtype=torch.float64)
    acts, rews, logged = [], [], []
    for _ in range(3):
        env.train_mode = True
        ob, done = env.reset(), True
        model.reset()
        n0 = len(tr.log)
        while True:
            ob, done, R = tr.explore(ob, done)
            model.backward(R, 0)
            if done:
                break
        acts += [a for a, _ in tr.log[n0:]]
        rews += [g for _, g in tr.log[n0:]]
        env.train_mode = False
        m, s, n = pfs.perform(tr)
        logged.append((m, s))
    train = z['train']
    np.testing.assert_array_equal(np.array(acts), z['actions'][train])
    np.testing.assert_allclose(np.array(rews), z['rewards'][train], rtol=1e-9)
    np.testing.assert_allclose(np.array(logged), z['logged'][:, :2], rtol=1e-9)


def test_product_reference_api_replays_three_episodes_on_cpu_emulation():
    from cpu_emulation import cpu_ops
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.utils import Counter, Trainer
    from oracle import trainer_ref
    z, cp = _golden()
    env = trainer_ref.RefCaccEnv(cp['ENV_CONFIG'])               # seeds np.random like CACCEnv.__init__
    env.terminate = lambda: None
    log = {'a': [], 'g': [], 'train': []}
    orig_step, orig_reset = env.step, env.reset
    n_train = [0]
    counter = Counter(10 ** 9, 10 ** 9, 10 ** 9)

    def reset(gui=False, test_ind=-1):
        if env.train_mode:
            n_train[0] += 1
            counter.stop = n_train[0] == 3
        return orig_reset(test_ind=test_ind)

    def step(action):
        out = orig_step(action)
        log['a'].append(np.array(action).copy()); log['g'].append(out[3]); log['train'].append(env.train_mode)
        return out
    env.step, env.reset = step, reset
    with cpu_ops():
        model = models.MA2C_NC(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                               cp['MODEL_CONFIG'], seed=int(z['seed']), num_envs=1, device='cpu')
        tr = Trainer(env, model, counter, None, output_path=None)
        tr.run()
    acts = np.array(log['a'])
    n = min(len(acts), len(z['actions']))
    same = np.all(acts[:n] == z['actions'][:n], axis=1)
    first_div = n if same.all() else int(np.argmin(same))
    # fp32 product vs the float64 reference: a draw can flip where a uniform falls within rounding of a CDF boundary
    assert first_div >= 420, 'diverged at step %d (episode %d)' % (first_div, z['episode'][first_div])    # well into episode 2
    np.testing.assert_allclose(np.array(log['g'])[:first_div], z['rewards'][:first_div], rtol=1e-4, atol=1e-2)
    if first_div == len(z['actions']) == len(acts):
        np.testing.assert_allclose([[d['avg_reward'], d['std_reward'], d['step']] for d in tr.data], z['logged'], rtol=1e-3)
        s = var_stats_from_named(model.policy.params.ref_variables())
        np.testing.assert_allclose(s[:, 1:3], z['stats'][:, 1:3], rtol=2e-3, atol=2e-5)
