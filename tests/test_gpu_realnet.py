"""GPU parity of the synthetic-network HIP kernel (through the C-ABI) against its specification
oracle/realnet_ref.py, and the heterogeneous nets trained on it by the batched engine."""
import numpy as np
import pytest
import torch

from helpers import net_config

pytestmark = pytest.mark.gpu


def make(E, coop_gamma=0.9, env_id_base=0, seed=12, agent='ma2c_nc'):
    from deeprl_network_amd.envs.real_net_env import RealNetBatchEnv
    return RealNetBatchEnv(net_config(agent=agent, coop_gamma=coop_gamma, seed=seed)['ENV_CONFIG'], num_envs=E,
                           env_id_base=env_id_base)


def rand_actions(rng, E, tp):
    return np.stack([rng.randint(0, tp.n_a_ls[i], size=E) for i in range(tp.N)], axis=1)


@pytest.mark.parametrize('E', [1, 7, 8, 9, 300])
@pytest.mark.parametrize('coop_gamma', [-1, 0.9])
def test_trajectory_vs_oracle(E, coop_gamma):
    from oracle import realnet_ref as R
    env = make(E, coop_gamma)
    tp = R.TOPO
    rng = np.random.RandomState(E)
    U = rng.rand(E, 4).astype(np.float32)
    env.reset(u0=torch.from_numpy(U).cuda())
    ref = R.NetBatchRef(R.NetParams(config=env.config), E=E, dtype=np.float32)
    ref.reset(np.float32(0.8) + np.float32(0.4) * U)
    np.testing.assert_array_equal(env.xi.cpu().numpy(), ref.xi)
    for t in range(200):
        hold = rng.rand(E, tp.N) < 0.6                    # keep the phase most of the time
        a = np.where(hold & (t > 0), ref.prev, rand_actions(rng, E, tp)).astype(np.uint8)
        obs, r, d, g = env.step(torch.from_numpy(a).cuda())
        ro, rr, rd, rg = ref.step(a)
        np.testing.assert_allclose(env.q.cpu().numpy(), ref.q, rtol=2e-4, atol=2e-3, err_msg='q t=%d' % t)
        np.testing.assert_allclose(env.transit.cpu().numpy(), ref.tr, rtol=2e-4, atol=2e-3, err_msg='tr t=%d' % t)
        np.testing.assert_allclose(obs.cpu().numpy(), R.gather_net(ro), rtol=2e-4, atol=1e-3)
        np.testing.assert_allclose(g.cpu().numpy(), rg, rtol=2e-4, atol=5e-2)
        np.testing.assert_allclose(r.cpu().numpy(), rr, rtol=2e-4, atol=5e-2)
        assert np.array_equal(d.cpu().numpy().astype(bool), rd)
        assert np.array_equal(env.prev_action.cpu().numpy(), a)


def test_single_step_tight_from_random_state():
    """One step from identical random states (no accumulated drift): rtol 1e-5."""
    from oracle import realnet_ref as R
    E, tp = 512, R.TOPO
    env = make(E)
    rng = np.random.RandomState(3)
    env.reset(u0=torch.from_numpy(rng.rand(E, 4).astype(np.float32)).cuda())
    ref = R.NetBatchRef(R.NetParams(config=env.config), E=E, dtype=np.float32)
    ref.reset(env.xi.cpu().numpy())
    ref.q = (rng.uniform(0, 30, size=(E, tp.N, tp.L)) * (rng.rand(E, tp.N, tp.L) < 0.8) * ref.valid).astype(np.float32)
    ref.q = np.minimum(ref.q, np.float32(26.0))
    ref.tr = (rng.uniform(0, 3, size=(E, tp.N, tp.L)) * ref.valid).astype(np.float32)
    ref.prev = rand_actions(rng, E, tp)
    ref.t = rng.randint(0, 700, size=E)
    env.q.copy_(torch.from_numpy(ref.q)); env.transit.copy_(torch.from_numpy(ref.tr))
    env.prev_action.copy_(torch.from_numpy(ref.prev.astype(np.uint8))); env.t.copy_(torch.from_numpy(ref.t.astype(np.int32)))
    a = rand_actions(rng, E, tp).astype(np.uint8)
    obs, r, d, g = env.step(torch.from_numpy(a).cuda())
    ro, rr, rd, rg = ref.step(a)
    np.testing.assert_allclose(env.q.cpu().numpy(), ref.q, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(env.transit.cpu().numpy(), ref.tr, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(obs.cpu().numpy(), R.gather_net(ro), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r.cpu().numpy(), rr, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(g.cpu().numpy(), rg, rtol=1e-5, atol=2e-3)


def test_episode_end_auto_reset_and_philox():
    from oracle import philox
    E, base, seed = 40, 1000, 77
    env = make(E, env_id_base=base, seed=seed)
    env.reset()
    U0 = np.stack(philox.philox4x32(base + np.arange(E), 0, 0, 0, seed, 0), axis=-1)
    np.testing.assert_array_equal(env.xi.cpu().numpy(), np.float32(0.8) + np.float32(0.4) * philox.u01(U0))
    env.t.fill_(env.T - 1)
    a = torch.zeros(E, env.n_agent, dtype=torch.uint8, device='cuda')
    obs, r, d, g = env.step(a, auto_reset=True)
    assert d.all() and (env.t == 0).all() and (env.q == 0).all() and (env.transit == 0).all() and (obs == 0).all()
    assert (env.episode == 2).all()
    U1 = np.stack(philox.philox4x32(base + np.arange(E), 0, 1, 0, seed, 0), axis=-1)
    np.testing.assert_array_equal(env.xi.cpu().numpy(), np.float32(0.8) + np.float32(0.4) * philox.u01(U1))


def test_reference_api_env_returns_ragged_lists():
    from deeprl_network_amd.envs import init_env
    for agent in ('ma2c_nc', 'ia2c_fp'):
        env = init_env(net_config(agent=agent)['ENV_CONFIG'])
        ob = env.reset()
        assert [len(o) for o in ob] == [env.n_s_ls[i] + (sum(env.n_a_ls[j] for j in env._nbr[i]) if agent == 'ia2c_fp' else 0)
                                        for i in range(env.n_agent)]
        a = [np.random.randint(0, n) for n in env.n_a_ls]
        ob, r, d, g = env.step(a)
        assert np.asarray(r).shape == (env.n_agent,) and not d and g <= 0
        assert [len(x) for x in env.get_neighbor_action(a)] == [len(js) for js in env._nbr]


@pytest.mark.parametrize('agent', ['ma2c_nc', 'ia2c_fp', 'ma2c_ic3'])
def test_heterogeneous_nets_train_on_the_network(agent):
    """The batched engine (hipGraph rollout + update) with 28 heterogeneous agents: finite, deterministic, actions
    inside every agent's own action set, padded parameters untouched."""
    from deeprl_network_amd.main import AGENTS
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    outs = []
    for rep in range(2):
        cp = net_config(agent=agent, n_step=24)
        env = make(64, agent=agent)
        np.random.seed(5)
        model = AGENTS[agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                              cp['MODEL_CONFIG'], seed=5, num_envs=64, device='cuda', n_feat_ls=env.n_feat_ls)
        assert not model.identical_agent and model.policy.hetero
        pad0 = model.policy.params['pi_b'].detach().clone()
        tr = BatchedTrainer(env, model, Counter(10 ** 9, 10 ** 9, 10 ** 9), use_graph=True)
        for _ in range(4):
            tr.run_batch()
        torch.cuda.synchronize()
        acts = model.buf_act.cpu().numpy()
        for i, n in enumerate(env.n_a_ls):
            assert acts[:, :, i].max() < n
        flat = model.policy.params.flat
        assert torch.isfinite(flat).all()
        pb = model.policy.params['pi_b'].detach()
        for i, n in enumerate(env.n_a_ls):
            assert torch.equal(pb[i, n:], pad0[i, n:]) and (pb[i, n:] < -1e29).all()
        if model.policy.params.mask is not None:
            frozen = model.policy.params.mask == 0
            assert (model.policy.params.grad[frozen] == 0).all()
        outs.append(flat.clone())
    assert torch.equal(outs[0], outs[1])
