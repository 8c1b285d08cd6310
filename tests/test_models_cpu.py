"""Host-logic parity of agents.models / agents.policies against the golden vectors produced by
the REAL reference model code on the fake-TF shim (tests/golden/nn_*.npz).  The HIP ops are
replaced by their oracle restatements (tests/cpu_emulation.py) because this container has no GPU;
tests/test_gpu_models.py runs the same comparison through the real kernels."""
import glob
import os

import numpy as np
import pytest
import torch

from cpu_emulation import cpu_ops
from helpers import GOLDEN, build_product_model, compare_scripted, drive_scripted, load_npz, var_stats_from_named

CASES = sorted(glob.glob(os.path.join(GOLDEN, 'nn_*.npz')))


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(c)[3:-4] for c in CASES])
def test_scripted_run_matches_reference(path):
    z = load_npz(path)
    with cpu_ops():
        model = build_product_model(z, 'cpu')
        named = model.policy.params.ref_variables()
        # initial weights: identical np.random draw order and shapes as the reference
        assert [n for n, _ in named] == [str(n) for n in z['names']]
        assert [str(a.shape) for _, a in named] == [str(s) for s in z['shapes']]
        np.testing.assert_allclose(var_stats_from_named(named), z['stats0'], rtol=1e-6, atol=1e-7)
        out = drive_scripted(model, z)
    compare_scripted(out, z)


def test_ortho_init_matches_reference():
    from deeprl_network_amd.agents.policies import ortho_init
    z = load_npz(os.path.join(GOLDEN, 'ortho_init.npz'))
    np.random.seed(12)
    for k, s in enumerate(z['shapes']):
        w = ortho_init(tuple(int(x) for x in s))
        assert w.dtype == np.float32 and np.array_equal(w, z['w%d' % k])
    assert np.random.rand() == float(z['after'])


@pytest.mark.parametrize('name', ['ma2c_nc_ragged', 'ia2c_fp_ragged', 'ma2c_ic3_ragged'])
def test_heterogeneous_checkpoint_roundtrip_and_layout(name, tmp_path):
    """Heterogeneous nets: variables are exported / re-imported under the reference's ragged shapes, padded entries
    keep their fill (0, -1e30 for absent actions), and the update mask covers exactly the exported entries."""
    import torch
    z = load_npz(os.path.join(GOLDEN, 'nn_%s.npz' % name))
    with cpu_ops():
        m1 = build_product_model(z, 'cpu')
        ps = m1.policy.params
        named = ps.ref_variables()
        assert [str(tuple(a.shape)) for _, a in named] == [str(s) for s in z['shapes']]
        n_exported = sum(a.size for _, a in named)
        if ps.mask is not None:
            # mask == 1 exactly on entries of existing variables; padded-but-existing tensors are larger than the export
            assert int(ps.mask.sum().item()) == n_exported
        m1.save(str(tmp_path) + '/', 7)
        m2 = build_product_model(z, 'cpu')
        with torch.no_grad():
            m2.policy.params.flat.add_(0.5)              # scramble, incl. the padding
        assert m2.load(str(tmp_path) + '/')
        for key in m1.policy.params.index:                   # every tensor incl. its padded entries (alignment gaps aside)
            assert torch.equal(m1.policy.params[key], m2.policy.params[key]), key
        pb = m2.policy.params['pi_b'].detach()
        for i, n in enumerate(m2.n_a_ls):
            assert (pb[i, n:] == -1e30).all() and (pb[i, :n] == 0).all()


BATCHED = sorted(glob.glob(os.path.join(GOLDEN, 'nnb_*.npz')))


@pytest.mark.parametrize('saved', [False, True], ids=['recompute', 'saved_acts'])
@pytest.mark.parametrize('path', BATCHED, ids=[os.path.basename(c)[4:-4] for c in BATCHED])
def test_batched_update_equals_mean_of_reference_replica_gradients(path, saved):
    """E = K = 4 replicas, n_step 60 / 120: the product's batched update == mean over K independent reference models'
    gradients -> clip -> one RMSProp step (tests/golden/make_golden_nn.py run_batched), host logic on CPU emulation."""
    from helpers import build_product_batched, compare_batched, drive_batched
    z = load_npz(path)
    with cpu_ops():
        model = build_product_batched(z, 'cpu')
        if saved and not model.policy.can_save_acts:
            pytest.skip('coupled net: the update recomputes its forward pass')
        np.testing.assert_allclose(var_stats_from_named(model.policy.params.ref_variables()), z['stats0'], rtol=1e-6, atol=1e-7)
        out = drive_batched(model, z, saved=saved)
    compare_batched(out, z)


def test_ic3_encoder_inside_the_step_equals_separate_encoder():
    """CommNet on a compact observation with 16-byte feature pieces (the grid's shape): step_policy_value with `ob` (the
    one-launch step runs the observation encoder itself and writes its output to the enc slot) == encode() followed by
    the step on the encoder's output -- the host side of the in-kernel encoder (policies.IC3MultiAgentPolicy._ob_spec)."""
    from deeprl_network_amd import ops
    from deeprl_network_amd.agents.policies import IC3MultiAgentPolicy
    with cpu_ops():
        N, E, F, A = 6, 5, 8, 4
        mask = np.zeros((N, N), dtype=int)
        for i in range(N - 1):                                  # a line: ragged neighbour counts (1 or 2)
            mask[i, i + 1] = mask[i + 1, i] = 1
        np.random.seed(3)
        pol = IC3MultiAgentPolicy(F, A, mask, device='cpu')
        pol.params.init_reference_order()
        pol.refresh_wimage()
        assert pol.encodes_in_step(E, True) and not pol.encodes_in_step(E, False)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(E, N, F, generator=g)
        h, c = torch.randn(N, E, 64, generator=g) * 0.5, torch.randn(N, E, 64, generator=g) * 0.5
        done = (torch.rand(E, generator=g) < 0.4).float()
        outs = []
        for in_step in (True, False):
            enc = torch.full((N, E, 64), 3.0)
            if not in_step:
                pol.encode(x, None, out=enc)
            pi, act, v = torch.zeros(N, E, A), torch.zeros(E, N, dtype=torch.uint8), torch.zeros(N, E)
            ho, co, gates, S = torch.zeros_like(h), torch.zeros_like(c), torch.zeros(N, E, 256), torch.zeros(N, E, 64)
            pol.step_policy_value(enc, h, c, done, pi, act, v, h_out=ho, c_out=co, gates=gates, defer_action_term=True,
                                  save={'S': S}, mode=ops.SAMPLE_PHILOX, seed=4, env_id_base=0, step=2,
                                  **(dict(ob=x) if in_step else {}))
            outs.append((enc, pi, act, v, ho, co, gates, S))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
