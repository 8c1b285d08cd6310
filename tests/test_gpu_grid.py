"""GPU parity of the synthetic-grid HIP kernel (through the C-ABI) against its specification
oracle/grid_ref.py, plus properties at the BASELINE size (E = 1024)."""
import numpy as np
import pytest
import torch

from helpers import grid_config

pytestmark = pytest.mark.gpu


def make(E, coop_gamma=-1, env_id_base=0, seed=12):
    from deeprl_network_amd.envs.large_grid_env import LargeGridBatchEnv
    return LargeGridBatchEnv(grid_config(coop_gamma=coop_gamma, seed=seed)['ENV_CONFIG'], num_envs=E,
                             env_id_base=env_id_base)


@pytest.mark.parametrize('E', [1, 7, 8, 9, 300])
@pytest.mark.parametrize('coop_gamma', [-1, 0.9])
def test_trajectory_vs_oracle(E, coop_gamma):
    from oracle import grid_ref as G
    env = make(E, coop_gamma)
    rng = np.random.RandomState(E)
    U = rng.rand(E, 4).astype(np.float32)
    env.reset(u0=torch.from_numpy(U).cuda())
    ref = G.GridBatchRef(G.GridParams(config=env.config), E=E, dtype=np.float32)
    ref.reset(np.float32(0.8) + np.float32(0.4) * U)
    np.testing.assert_array_equal(env.xi.cpu().numpy(), ref.xi)
    for t in range(150):
        hold = rng.rand(E, 25) < 0.6                      # keep the phase most of the time
        a = np.where(hold & (t > 0), ref.prev, rng.randint(0, 5, size=(E, 25))).astype(np.uint8)
        obs, r, d, g = env.step(torch.from_numpy(a).cuda())
        ro, rr, rd, rg = ref.step(a)
        np.testing.assert_allclose(env.q.cpu().numpy(), ref.q, rtol=2e-4, atol=2e-3, err_msg='q t=%d' % t)
        np.testing.assert_allclose(env.transit.cpu().numpy(), ref.tr, rtol=2e-4, atol=2e-3, err_msg='tr t=%d' % t)
        np.testing.assert_allclose(obs.cpu().numpy(), G.gather_grid(ro), rtol=2e-4, atol=1e-3)
        np.testing.assert_allclose(g.cpu().numpy(), rg, rtol=2e-4, atol=2e-2)
        np.testing.assert_allclose(r.cpu().numpy(), rr, rtol=2e-4, atol=2e-2)
        assert np.array_equal(d.cpu().numpy().astype(bool), rd)
        assert np.array_equal(env.prev_action.cpu().numpy(), a)


def test_single_step_tight_from_random_state():
    """One step from identical random states (no accumulated drift): rtol 1e-5."""
    from oracle import grid_ref as G
    E = 512
    env = make(E)
    rng = np.random.RandomState(3)
    env.reset(u0=torch.from_numpy(rng.rand(E, 4).astype(np.float32)).cuda())
    ref = G.GridBatchRef(G.GridParams(config=env.config), E=E, dtype=np.float32)
    ref.reset(env.xi.cpu().numpy())
    ref.q = rng.uniform(0, 30, size=(E, 25, 6)).astype(np.float32) * (rng.rand(E, 25, 6) < 0.8)
    ref.tr = rng.uniform(0, 3, size=(E, 25, 6)).astype(np.float32)
    ref.prev = rng.randint(0, 5, size=(E, 25))
    ref.t = rng.randint(0, 700, size=E)
    env.q.copy_(torch.from_numpy(ref.q)); env.transit.copy_(torch.from_numpy(ref.tr))
    env.prev_action.copy_(torch.from_numpy(ref.prev.astype(np.uint8))); env.t.copy_(torch.from_numpy(ref.t.astype(np.int32)))
    a = rng.randint(0, 5, size=(E, 25)).astype(np.uint8)
    obs, r, d, g = env.step(torch.from_numpy(a).cuda())
    ro, rr, rd, rg = ref.step(a)
    np.testing.assert_allclose(env.q.cpu().numpy(), ref.q, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(env.transit.cpu().numpy(), ref.tr, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(obs.cpu().numpy(), G.gather_grid(ro), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g.cpu().numpy(), rg, rtol=1e-5, atol=1e-3)


def test_episode_end_auto_reset_and_philox():
    from oracle import philox
    E, base, seed = 64, 500, 12
    env = make(E, env_id_base=base, seed=seed)
    env.config  # noqa: B018
    env.params.T = 6                      # short episode
    env.reset()
    U0 = np.stack(philox.philox4x32(base + np.arange(E), 0, 0, 0, seed, 0), axis=-1)
    np.testing.assert_array_equal(env.xi.cpu().numpy(), np.float32(0.8) + np.float32(0.4) * philox.u01(U0))
    a = torch.zeros(E, 25, dtype=torch.uint8, device='cuda')
    for t in range(6):
        obs, r, d, g = env.step(a, auto_reset=True)
        assert bool(d.all()) == (t == 5)
    assert torch.all(env.t == 0) and torch.all(env.q == 0) and torch.all(env.obs == 0) and torch.all(env.episode == 2)
    U1 = np.stack(philox.philox4x32(base + np.arange(E), 0, 1, 0, seed, 0), axis=-1)
    np.testing.assert_array_equal(env.xi.cpu().numpy(), np.float32(0.8) + np.float32(0.4) * philox.u01(U1))


def test_full_size_batch_invariance():
    E = 1024
    env = make(E)
    env.reset()
    small = make(8, env_id_base=400)
    small.reset()
    rng = np.random.RandomState(5)
    for t in range(40):
        a = torch.from_numpy(rng.randint(0, 5, size=(E, 25)).astype(np.uint8)).cuda()
        env.step(a)
        small.step(a[400:408].contiguous())
        assert torch.equal(env.q[400:408], small.q) and torch.equal(env.obs[400:408], small.obs)
    assert torch.isfinite(env.obs).all() and float(env.global_reward.max()) <= 0
