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


@pytest.mark.parametrize('E', [9, 300, 1024])
def test_compact_observation_is_the_own_wave_block(E):
    """p.compact_obs: obs [E,25,12] (every node's own wave vector -- what the reference hands an MA2C agent,
    atsc_env.py:253-262) == columns 0..11 of the gathered [E,25,60] slab; identical state, reward, done; the gathered
    slab is the neighbour gather of the compact one (ascending node index, zero padded)."""
    from deeprl_network_amd import ops
    a_env, b_env = make(E), make(E)
    assert b_env.set_compact_obs(True) and b_env.obs.shape == (E, 25, 12)
    a_env.reset(); b_env.reset()
    nbr_idx, _ = ops.neighbor_table(a_env.neighbor_mask, 'cuda')
    rng = np.random.RandomState(E)
    for t in range(60):
        a = torch.from_numpy(rng.randint(0, 5, size=(E, 25)).astype(np.uint8)).cuda()
        oa, ra, da, ga = a_env.step(a)
        ob, rb, db, gb = b_env.step(a)
        assert torch.equal(oa[:, :, :12], ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ga, gb)
        assert torch.equal(a_env.q, b_env.q) and torch.equal(a_env.transit, b_env.transit)
        own = ob.transpose(0, 1).contiguous()                                  # [25,E,12]
        torch.testing.assert_close(oa[:, :, 12:], ops.nbr_gather(own, nbr_idx).transpose(0, 1), rtol=0, atol=0)


def test_compact_observation_training_equals_gathered_slab():
    """Grid CommNet (BASELINE configs[3] at E = 256): the env writing compact observations -- the one-launch lock-step then
    runs the observation encoder itself on the matrix cores (lstm_step_x_kernel<4,2>, OBENC), the update's encoder backward
    gathers the neighbours inside its kernel -- == the env writing the gathered slab and a separate encoder launch.  The two
    encoders sum in different orders (last-bit differences in enc), so a sampled action can flip: replicas whose actions agree
    over the whole batch (required: > 99 %) must agree in their values and saved activations; weights after the update close."""
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.large_grid_env import LargeGridBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    out = []
    for compact in (True, False):
        cp = grid_config()
        env = LargeGridBatchEnv(cp['ENV_CONFIG'], num_envs=256)
        np.random.seed(12)
        model = models.MA2C_IC3(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                                cp['MODEL_CONFIG'], seed=12, num_envs=256)
        tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True, compact_obs=compact)
        assert tr.compact_obs == compact and model.buf_x.shape[-1] == (12 if compact else 60)
        tr.run_batch()
        torch.cuda.synchronize()
        assert model.policy.encodes_in_step(256, model.compact_obs) == compact
        out.append((model.policy.params.flat.clone(), model.buf_v.clone(), model.buf_act.clone(), model.policy._extra['ENC'].clone(),
                    model.S_buf.clone()))
        del env, model, tr
    same = (out[0][2] == out[1][2]).all(dim=2).all(dim=0)                    # [E]: replicas with identical action sequences
    assert same.float().mean() > 0.99, float(same.float().mean())
    torch.testing.assert_close(out[0][1][:, :, same], out[1][1][:, :, same], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out[0][3][:, :, same], out[1][3][:, :, same], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(out[0][4][:, :, same], out[1][4][:, :, same], rtol=1e-4, atol=2e-5)
    if bool(same.all()):
        torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize('objective,coop_gamma,E', [('wait', -1, 9), ('hybrid', 0.9, 300), ('hybrid', -1, 8)])
def test_wait_and_hybrid_objectives_vs_oracle(objective, coop_gamma, E):
    """atsc_env.py:383-418's `wait` / `hybrid` objectives on the synthetic grid: the per-lane front-vehicle standing time
    (oracle/grid_ref.py step 6) as one more state array; trajectories, rewards and the state itself against the oracle,
    incl. the fused auto-reset clearing it."""
    from oracle import grid_ref as G
    from deeprl_network_amd.envs.large_grid_env import LargeGridBatchEnv
    cp = grid_config(coop_gamma=coop_gamma)
    cp['ENV_CONFIG']['objective'] = objective
    cp['ENV_CONFIG']['coef_wait'] = '0.2'
    env = LargeGridBatchEnv(cp['ENV_CONFIG'], num_envs=E)
    assert env.head_wait is not None and env.head_wait in [t for t in env.state_tensors() if t is env.head_wait]
    rng = np.random.RandomState(E)
    U = rng.rand(E, 4).astype(np.float32)
    env.reset(u0=torch.from_numpy(U).cuda())
    ref = G.GridBatchRef(G.GridParams(config=env.config), E=E, dtype=np.float32)
    assert ref.p.objective == objective and ref.p.coef_wait == pytest.approx(0.2)
    ref.reset(np.float32(0.8) + np.float32(0.4) * U)
    seen_wait = 0.0
    for t in range(120):
        hold = rng.rand(E, 25) < 0.7
        a = np.where(hold & (t > 0), ref.prev, rng.randint(0, 5, size=(E, 25))).astype(np.uint8)
        # the standing time is a threshold decision on queues that drift by fp32 rounding between the two free-running
        # trajectories: it is handed over from the oracle before every step (so one flipped decision does not persist) and
        # compared exactly, a lane whose discharge / queue sits within rounding of the threshold excepted
        env.head_wait.copy_(torch.from_numpy(ref.hw))
        q_before = ref.q.copy()
        obs, r, d, g = env.step(torch.from_numpy(a).cuda())
        ro, rr, rd, rg = ref.step(a)
        bad = env.head_wait.cpu().numpy() != ref.hw
        assert bad.mean() < 2e-3, 'head_wait differs in %d lanes at t=%d' % (bad.sum(), t)
        np.testing.assert_allclose(env.q.cpu().numpy(), ref.q, rtol=2e-4, atol=2e-3)
        ok = ~bad.any(axis=(1, 2))                         # replicas without a flipped threshold: rewards as usual
        np.testing.assert_allclose(g.cpu().numpy()[ok], rg[ok], rtol=2e-4, atol=5e-2)
        np.testing.assert_allclose(r.cpu().numpy()[ok], rr[ok], rtol=2e-4, atol=5e-2)
        del q_before
        seen_wait = max(seen_wait, float(ref.hw.max()))
    assert seen_wait >= 10.0                                   # queues did stand through several red steps
    # the last step of an episode with the fused auto-reset clears the state
    env.t.fill_(env.T - 1)
    env.step(torch.zeros(E, 25, dtype=torch.uint8, device='cuda'), auto_reset=True)
    assert torch.all(env.head_wait == 0) and torch.all(env.q == 0)
    # and the `queue` instantiation is untouched by the new state
    assert make(4).head_wait is None


def test_reference_api_ia2c_observation_order_and_greedy_evaluate(tmp_path):
    """E = 1 duck-type: an IA2C agent's observation lists its neighbours north, east, south, west (atsc_env.py:263-271), the
    IA2C-FP one appends their fingerprints in that order; and `main.py evaluate` on an `agent = greedy` run directory drives the
    rule-based controller (large_grid_env.py:30-45) through Evaluator.perform."""
    from deeprl_network_amd.envs.large_grid_env import LargeGridEnv, grid_neighbor_order
    order = grid_neighbor_order()
    cp = grid_config(agent='ia2c_fp')
    env = LargeGridEnv(cp['ENV_CONFIG'])
    env.train_mode = True
    ob = env.reset()
    assert [len(o) for o in ob] == [12 * (1 + len(order[i])) + 5 * len(order[i]) for i in range(25)]
    rng = np.random.RandomState(0)
    for _ in range(30):
        ob, r, d, g = env.step(rng.randint(0, 5, size=25))
    env.update_fingerprint([rng.dirichlet(np.ones(5)) for _ in range(25)])
    ob = env._state_list()
    own = env.batch.obs[0, :, :12].cpu().numpy()
    assert own.max() > 0
    for i in (0, 4, 7, 12, 24):
        want = np.concatenate([own[i]] + [own[j] for j in order[i]] + [env.fp[j] for j in order[i]])
        np.testing.assert_allclose(ob[i], want, rtol=0, atol=0)
    # greedy evaluation through the CLI
    import shutil
    import pandas as pd
    from deeprl_network_amd.main import main
    base = tmp_path / 'greedy'
    (base / 'data').mkdir(parents=True)
    (base / 'model').mkdir()
    cp = grid_config(agent='greedy', coop_gamma=0.75)
    cp['ENV_CONFIG']['episode_length_sec'] = '300'
    with open(base / 'data' / 'config_greedy.ini', 'w') as f:
        cp.write(f)
    main(['--base-dir', str(base), 'evaluate', '--evaluation-seeds', '10000,20000'])
    df = pd.read_csv(str(base / 'eva_data') + '/atsc_large_grid_greedy_control.csv')
    assert len(df) == 2 * 60 and set(df['episode']) == {1, 2} and (df['reward'] <= 0).all()
    shutil.rmtree(base)
