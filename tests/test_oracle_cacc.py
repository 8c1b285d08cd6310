"""Pin oracle/cacc_ref.py against the REAL reference env (tests/golden/cacc_*.npz,
made by tests/golden/make_golden_env.py from /root/reference/envs/cacc_env.py)."""
import glob
import os

import numpy as np
import pytest

from oracle.cacc_ref import CaccBatchRef, CaccParams, gather_line

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(glob.glob(os.path.join(GOLDEN, 'cacc_*.npz')))


def load(path):
    with np.load(path, allow_pickle=False) as f:
        z = {k: f[k] for k in f.files}   # NpzFile re-inflates on every [] access
    p = CaccParams(scenario='cacc_' + str(z['scenario']), agent=str(z['agent']),
                   seed=int(z['seed']), coop_gamma=float(z['coop_gamma']))
    return z, p


def test_have_cases():
    assert len(CASES) >= 12


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(c)[5:-4] for c in CASES])
def test_oracle_bit_exact_vs_reference(path):
    z, p = load(path)
    env = CaccBatchRef(p, E=1, dtype=np.float64, train_mode=bool(z['train_mode']))
    env.reset([float(z['U'])])
    agent = str(z['agent'])
    fps = z['fps']
    n_s = z['n_s']

    def check_obs(k):
        ob = env.ref_obs_list(agent, fp=fps[k][None], e=0)
        for i, o in enumerate(ob):
            assert len(o) == n_s[i]
            assert np.array_equal(o, z['obs'][k, i, :n_s[i]]), (k, i)

    assert np.array_equal(env.h[0], z['h'][0]) and np.array_equal(env.v[0], z['v'][0])
    check_obs(0)
    for k, a in enumerate(z['acts']):
        _, r, d, g = env.step(a[None])
        assert np.array_equal(env.h[0], z['h'][k + 1]), k
        assert np.array_equal(env.v[0], z['v'][k + 1]), k
        assert np.array_equal(env.u[0], z['u'][k + 1]), k
        assert g[0] == z['global_reward'][k], k
        assert np.array_equal(np.broadcast_to(r[0], (p.n_agent,)), z['reward'][k]), k
        assert bool(d[0]) == bool(z['done'][k]), k
        check_obs(k + 1)
    assert bool(d[0])
    nb = z['neighbor_mask']
    assert np.array_equal(env.neighbor_mask, nb) and np.array_equal(env.distance_mask, z['distance_mask'])


def test_survey_known_answers():
    """SURVEY.md section 8(c) values (catch-up, seed 12, actions [3,1,2,0,3,1,2,0] x3)."""
    np.random.seed(12)
    U = np.random.rand()
    assert U == 0.15416284237967237
    env = CaccBatchRef(CaccParams(scenario='cacc_catchup'), E=1)
    ob = env.reset([U])
    np.testing.assert_allclose(env.h[0, 0], 33.0832568476, atol=1e-9)
    np.testing.assert_allclose(ob[0, 0], [0, 0, 2, 0.6541628424, 0], atol=1e-9)
    a = np.array([[3, 1, 2, 0, 3, 1, 2, 0]])
    g = [env.step(a)[3][0] for _ in range(3)]
    np.testing.assert_allclose(g, [-171.5323408189, -170.7432897535, -169.4407866496], atol=1e-8)
    np.testing.assert_allclose(env.obs()[0, 0], [0.05, -0.15, 2, 0.6447878424, 1], atol=1e-9)


def test_batched_equals_single():
    """E replicas in one call == E independent single-replica runs."""
    p = CaccParams(scenario='cacc_slowdown')
    rng = np.random.RandomState(0)
    E, T = 5, 130
    U = rng.rand(E)
    acts = rng.randint(0, 4, size=(T, E, 8))
    big = CaccBatchRef(p, E=E)
    big.reset(U)
    singles = [CaccBatchRef(p, E=1) for _ in range(E)]
    for e, s in enumerate(singles):
        s.reset([U[e]])
    for t in range(T):
        ob, r, d, g = big.step(acts[t])
        for e, s in enumerate(singles):
            ob1, r1, d1, g1 = s.step(acts[t, e][None])
            assert np.array_equal(ob[e], ob1[0]) and g[e] == g1[0] and d[e] == d1[0]


def test_fp32_variant_close_to_fp64():
    z, p = load(os.path.join(GOLDEN, 'cacc_catchup_nc_const3.npz'))
    env = CaccBatchRef(p, E=1, dtype=np.float32)
    env.reset([float(z['U'])])
    for k, a in enumerate(z['acts']):
        env.step(a[None])
    assert env.h.dtype == np.float32
    np.testing.assert_allclose(env.h[0], z['h'][-1], atol=1e-3)
    np.testing.assert_allclose(env.v[0], z['v'][-1], atol=1e-3)


def test_gather_line_layout():
    x = np.arange(2 * 8 * 5, dtype=np.float64).reshape(2, 8, 5)
    y = gather_line(x)
    assert y.shape == (2, 8, 15)
    assert np.array_equal(y[:, 0, 5:10], x[:, 1]) and np.all(y[:, 0, 10:] == 0)
    assert np.array_equal(y[:, 3, 5:10], x[:, 2]) and np.array_equal(y[:, 3, 10:], x[:, 4])
    assert np.array_equal(y[:, 7, 5:10], x[:, 6]) and np.all(y[:, 7, 10:] == 0)
