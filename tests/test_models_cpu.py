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
