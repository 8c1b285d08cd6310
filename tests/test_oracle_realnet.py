"""The synthetic Monaco-like network: what is pinned to the reference (tables, masks, action / observation widths),
the independence of the two transcriptions (oracle vs product), and the properties of the specified dynamics."""
import os
import sys
import types

import numpy as np
import pytest

from helpers import net_config
from oracle.realnet_ref import DET_CAP, N_GROUP, Q_MAX, TOPO, NetBatchRef, NetParams, gather_net

REF = '/root/reference'


@pytest.mark.skipif(not os.path.exists(REF), reason='reference checkout not present (GPU box)')
def test_tables_and_masks_equal_the_reference():
    """NODES / PHASES (real_net_env.py:21-69) and the masks RealNetEnv builds from them (152-195), by importing the
    reference module itself (SUMO / plotting imports stubbed: only module-level constants and pure methods are used)."""
    saved = dict(sys.modules)
    try:
        for m in ('seaborn', 'matplotlib', 'matplotlib.pyplot', 'traci', 'traci.exceptions', 'sumolib'):
            sys.modules[m] = types.ModuleType(m)
        sys.modules['seaborn'].set_color_codes = lambda *a, **k: None
        sys.modules['sumolib'].checkBinary = lambda x: x
        sys.modules['traci'].exceptions = types.SimpleNamespace(FatalTraCIError=Exception)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'tf1_shim'))
        sys.path.insert(0, REF)
        for k in [k for k in sys.modules if k == 'envs' or k.startswith('envs.')]:
            del sys.modules[k]
        from envs.real_net_env import NODES, PHASES, RealNetEnv
        names = sorted(NODES)
        assert names == TOPO.names
        for i, n in enumerate(names):
            assert list(PHASES[NODES[n][0]]) == list(TOPO.phases[i])
            assert [names.index(x) for x in NODES[n][1]] == TOPO.nbrs_listed[i]
        env = RealNetEnv.__new__(RealNetEnv)                     # no SUMO: run only the pure map builders
        env.node_names, env.n_node = names, len(names)
        env._init_neighbor_map()
        env._init_distance_map()
        np.testing.assert_array_equal(env.neighbor_mask, TOPO.neighbor_mask)
        np.testing.assert_array_equal(env.distance_mask, TOPO.distance_mask)
    finally:
        sys.path[:] = [p for p in sys.path if p != REF and not p.endswith('tf1_shim')]
        for k in [k for k in sys.modules if k not in saved]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_shapes_of_the_heterogeneous_system():
    tp = TOPO
    assert (tp.N, tp.A, tp.L, tp.m_max) == (28, 6, 22, 4)
    assert min(tp.n_a_ls) == 2 and max(tp.n_a_ls) == 6 and min(tp.n_s_ls) == 2 and max(tp.n_s_ls) == 22
    assert sum(len(js) == 0 for js in tp.nbrs) == 4                      # 8996, 9433, 9480, cluster_8751_9630
    assert (tp.neighbor_mask != tp.neighbor_mask.T).any()                # the listed neighbourhoods are directed
    assert (tp.distance_mask < 0).any() and tp.distance_mask.max() == 8
    assert int((tp.src >= 0).sum()) + int((tp.group >= 0).sum()) == sum(tp.n_s_ls)
    np.testing.assert_allclose([tp.ext_share[tp.group == g].sum() for g in range(N_GROUP)], 1.0)


def test_product_topology_is_an_independent_transcription_of_the_same_network():
    """deeprl_network_amd/envs/real_net_env.py carries its own copy of the tables (the product never imports the
    oracle); both must describe the same network and the same derived link graph."""
    from deeprl_network_amd.envs.real_net_env import NetTopology
    pt = NetTopology('cpu')
    assert pt.node_names == TOPO.names and pt.n_a_ls == TOPO.n_a_ls and pt.n_s_ls == TOPO.n_s_ls
    np.testing.assert_array_equal(pt.neighbor_mask, TOPO.neighbor_mask)
    np.testing.assert_array_equal(pt.distance_mask, TOPO.distance_mask)
    np.testing.assert_array_equal(pt.host['green'], TOPO.green)
    np.testing.assert_array_equal(pt.host['src'], TOPO.src)
    np.testing.assert_array_equal(pt.host['fan'], TOPO.fan)
    np.testing.assert_array_equal(pt.host['group'], TOPO.group)
    np.testing.assert_allclose(pt.host['ext_share'], TOPO.ext_share, rtol=1e-7)
    # fan-out lists: ascending (node, link) per feeder, consistent with src
    ptr, pair = pt.host['dn_ptr'], pt.host['dn_pair']
    for j in range(pt.N):
        items = [(int(v) >> 8, int(v) & 255) for v in pair[ptr[j]:ptr[j + 1]]]
        assert items == sorted(items) and all(TOPO.src[i, k] == j for i, k in items) and len(items) == TOPO.fan[j]


def test_vehicles_are_conserved_and_queues_bounded():
    p = NetParams(config=net_config()['ENV_CONFIG'])
    ref = NetBatchRef(p, E=4)
    rng = np.random.RandomState(1)
    ref.reset(0.8 + 0.4 * rng.rand(4, N_GROUP))
    tp = TOPO
    for t in range(300):
        before = ref.q.sum(axis=(1, 2)) + ref.tr.sum(axis=(1, 2))
        a = np.stack([rng.randint(0, tp.n_a_ls[i], size=4) for i in range(tp.N)], axis=1)
        q0, tr0 = ref.q.copy(), ref.tr.copy()
        ob, r, d, g = ref.step(a)
        ext = ref.tr[:, tp.group >= 0].sum(axis=1) - 0.0          # arrivals of this step sit in transit of the entries
        fed_in = ref.tr[:, tp.src >= 0].sum(axis=1)
        out_of_net = before + ext + fed_in - (ref.q.sum(axis=(1, 2)) + ref.tr.sum(axis=(1, 2)))   # = served - accepted + ...
        # served vehicles either were accepted downstream (fed_in) or left through nodes that feed nothing
        served = (q0 + tr0 - ref.q).sum(axis=(1, 2))
        left = served - fed_in
        assert (left > -1e-9).all()
        leaves = np.array(tp.fan) == 0
        assert (ref.q <= Q_MAX + 1e-9).all() and (ref.q >= -1e-12).all()
        assert (ob <= DET_CAP / p.norm_wave + 1e-12).all() and ob.shape == (4, tp.N, tp.L)
        assert (ob[:, ~ref.valid] == 0).all()
        del out_of_net, leaves
    assert np.isfinite(r).all() and r.shape == (4, tp.N) and (g <= 0).all()


def test_yellow_logic_and_gather_layout():
    p = NetParams(config=net_config()['ENV_CONFIG'])
    ref = NetBatchRef(p, E=1)
    ref.reset(np.ones((1, N_GROUP)))
    tp = TOPO
    prev = np.zeros((1, tp.N), dtype=np.int64)
    cur = np.ones((1, tp.N), dtype=np.int64)
    g = ref._eff_green(prev, cur)
    i = tp.names.index('9429')                                    # phase set 5.0: 'GGGGg...' -> 'grrrG...'
    np.testing.assert_allclose(g[0, i, :5], [2.5, 1.0, 1.0, 1.0, 5.0])   # G->g: 5 s at half rate; G->r: 1 s; g->G: 5 s
    x = np.arange(tp.N * tp.L, dtype=np.float64).reshape(1, tp.N, tp.L)
    y = gather_net(x)
    j = tp.nbrs[i][0]
    np.testing.assert_array_equal(y[0, i, tp.L:2 * tp.L], x[0, j])
    np.testing.assert_array_equal(y[0, tp.names.index('8996'), tp.L:], 0)     # no listed neighbours
