"""The synthetic grid oracle: what IS pinned against the reference (topology, phases, demand schedule,
contract constants) and self-consistency of the specified dynamics (vehicle conservation, batching)."""
import os
import sys

import numpy as np
import pytest

from oracle import grid_ref as G

REF = '/root/reference'


def test_masks_match_reference_formulas():
    nb, dist = G.grid_masks()
    # large_grid_env.py:94-105 builds distance_mask from these blocks
    block0 = np.array([[0, 1, 2, 3, 4], [1, 0, 1, 2, 3], [2, 1, 0, 1, 2], [3, 2, 1, 0, 1], [4, 3, 2, 1, 0]])
    rows = [np.hstack([block0 + abs(i - j) for j in range(5)]) for i in range(5)]
    assert np.array_equal(dist, np.vstack(rows))
    # large_grid_env.py:58-92: corner 2, edge 3, internal 4 neighbours, lattice adjacency
    assert sorted(set(nb.sum(1))) == [2, 3, 4] and nb.sum() == 2 * 40
    assert list(np.where(nb[0])[0]) == [1, 5] and list(np.where(nb[12])[0]) == [7, 11, 13, 17]


@pytest.mark.skipif(not os.path.exists(REF), reason='reference checkout not present')
def test_phases_and_demand_pinned_to_reference_sources():
    src = open(os.path.join(REF, 'envs/large_grid_env.py')).read()
    for p in G.PHASES:
        assert "'%s'" % p in src
    bf = open(os.path.join(REF, 'envs/large_grid_data/build_file.py')).read()
    assert 'ratios1 = np.array([0.4, 0.7, 0.9, 1.0, 0.75, 0.5, 0.25])' in bf
    assert 'ratios2 = np.array([0.3, 0.8, 0.9, 1.0, 0.8, 0.6, 0.2])' in bf
    sys.path.insert(0, os.path.join(REF, 'envs', 'large_grid_data'))
    import build_file
    # entries: srcs of the 4 groups (build_file.py:285-289) map to (node, approach)
    srcs = [build_file.get_external_od(x, dest=False) for x in ([12, 13, 14], [16, 18, 20], [2, 3, 4], [6, 8, 10])]
    want = []
    for g, edges in enumerate(srcs):
        for e in edges:
            node = int(e.split('_')[1][2:]) - 1
            want.append((node, g))
    assert [(n, g) for n, _, g in G.ENTRIES] == want
    # schedule: the reference emits flows for pieces i < 7 (groups 0,1) and i >= 3 (groups 2,3)
    assert G.demand_rate(1, 0, 1100, 925) == 1100 * 0.4 and G.demand_rate(0, 2099, 1100, 925) == 1100 * .6 * .25
    assert G.demand_rate(0, 2100, 1100, 925) == 0 and G.demand_rate(2, 899, 1100, 925) == 0
    assert G.demand_rate(3, 900, 1100, 925) == 925 * 0.3 and G.demand_rate(3, 2999, 1100, 925) == 925 * 0.2
    assert G.demand_rate(3, 3000, 1100, 925) == 0


def test_link_tables_are_consistent():
    # every link's destination approach lists that link among its feeders, and vice versa
    for k, (dr, dc, ap) in enumerate(G.LINK_DEST):
        assert (-dr, -dc) == G.APPROACH_FROM[ap] and k in G.APPROACH_FEED[ap]
    assert sorted(k for f in G.APPROACH_FEED for k in f) == list(range(12))
    # shares of each physical lane sum to 1
    for lane in range(6):
        assert abs(G.LINK_SHARE[G.LINK_LANE == lane].sum() - 1) < 1e-12
    gt = G.green_table()
    assert gt.shape == (5, 12) and (gt[0] == [1, 1, 2, 0, 0, 0, 1, 1, 2, 0, 0, 0]).all()


def test_vehicle_conservation_and_bounds():
    p = G.GridParams()
    E = 4
    env = G.GridBatchRef(p, E=E)
    rng = np.random.RandomState(1)
    env.reset(0.8 + 0.4 * rng.rand(E, 4))
    entered = np.zeros(E)
    for t in range(300):
        sec = t * 5
        arr = sum(G.demand_rate(g, sec, p.peak1, p.peak2) / 3600 * 5 * env.xi[:, g] for _, _, g in G.ENTRIES)
        before = env.q.sum((1, 2)) + env.tr.sum((1, 2))
        D_exit_cap = before + arr
        ob, r, d, g = env.step(rng.randint(0, 5, size=(E, 25)))
        after = env.q.sum((1, 2)) + env.tr.sum((1, 2))
        assert np.all(after <= D_exit_cap + 1e-9)             # nothing is created
        assert np.all(env.q >= -1e-9) and np.all(env.tr >= -1e-9)
        assert np.all(ob >= 0) and np.all(ob <= 1.4 + 1e-12)   # min(7/5, clip 2)
        assert np.all(g <= 0) and np.all(g >= -25 * 12 * 7)
    assert not d.any()
    # internal lanes never exceed their storage (spill-back); entry lanes may queue outside
    internal = np.ones((25, 6), bool)
    for n, ap, _ in G.ENTRIES:
        internal[n, G.LANE_APPROACH == ap] = False
    assert env.q[:, internal].max() <= G.Q_MAX + 1e-9


def test_batch_equals_single_and_done_at_T():
    p = G.GridParams(episode_length_sec=100)
    assert p.T == 20
    rng = np.random.RandomState(2)
    xi = 0.8 + 0.4 * rng.rand(3, 4)
    acts = rng.randint(0, 5, size=(20, 3, 25))
    big = G.GridBatchRef(p, E=3)
    big.reset(xi)
    singles = [G.GridBatchRef(p, E=1) for _ in range(3)]
    for e, s in enumerate(singles):
        s.reset(xi[e:e + 1])
    for t in range(20):
        ob, r, d, g = big.step(acts[t])
        for e, s in enumerate(singles):
            ob1, r1, d1, g1 = s.step(acts[t, e:e + 1])
            assert np.array_equal(ob[e], ob1[0]) and g[e] == g1[0]
        assert d.all() == (t == 19)
    y = G.gather_grid(ob)
    assert y.shape == (3, 25, 60) and np.array_equal(y[:, 12, 12:24], ob[:, 7]) and np.all(y[:, 0, 36:] == 0)


def test_wait_and_hybrid_objectives_of_the_spec():
    """oracle/grid_ref.py step 6 (`wait` / `hybrid`, atsc_env.py:383-418): the front vehicle of a lane stands while a queue
    discharges nothing, any served flow or an empty lane clears it; rewards combine as the reference's three branches."""
    from oracle import grid_ref as G
    E = 3
    a_red = np.full((E, 25), 3)                           # phase 3 serves the E approach only
    refs = {o: G.GridBatchRef(G.GridParams(objective=o, coef_wait=0.5), E=E) for o in ('queue', 'wait', 'hybrid')}
    for r in refs.values():
        r.reset(np.ones((E, 4)))
        r.q[:] = 4.0                                      # standing queues everywhere
    out = {o: None for o in refs}
    for t in range(3):
        for o, r in refs.items():
            out[o] = r.step(a_red)
    hw = refs['wait'].hw
    assert np.all(hw[:, :, [4, 5]] == 15.0)               # three red steps of 5 s on the W lanes
    assert np.all(hw[:, :, [0, 3]] == 10.0)               # N / S: phase 0 -> 3 gives them 1 s of yellow clearance in the first step
    assert np.all(hw[:, :, [1, 2]] == 0.0)                # the served E lanes' front vehicles move
    wait = hw[:, :, G.LINK_LANE].sum(axis=2)
    np.testing.assert_allclose(out['wait'][1], -wait.sum(axis=1))
    np.testing.assert_allclose(out['hybrid'][1], out['queue'][1] - 0.5 * wait.sum(axis=1))
    np.testing.assert_array_equal(refs['queue'].q, refs['wait'].q)      # the objective changes the reward, not the traffic
    before = hw.copy()
    refs['wait'].reset(np.ones((E, 4)), mask=[1, 0, 0])
    assert np.all(refs['wait'].hw[0] == 0) and np.array_equal(refs['wait'].hw[1:], before[1:])
