"""The scripted three-batch golden comparison (reference model code on the fake-TF shim) through the
REAL HIP kernels + rocBLAS GEMMs on the GPU.  Tolerances: SURVEY.md 8(c)."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, build_product_model, compare_scripted, drive_scripted, load_npz, var_stats_from_named

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(GOLDEN, 'nn_*.npz')))


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(c)[3:-4] for c in CASES])
def test_scripted_run_matches_reference_on_gpu(path):
    z = load_npz(path)
    model = build_product_model(z, 'cuda')
    named = model.policy.params.ref_variables()
    np.testing.assert_allclose(var_stats_from_named(named), z['stats0'], rtol=1e-6, atol=1e-7)
    out = drive_scripted(model, z)
    compare_scripted(out, z)


BATCHED = sorted(glob.glob(os.path.join(GOLDEN, 'nnb_*.npz')))


@pytest.mark.parametrize('saved', [False, True], ids=['recompute', 'saved_acts'])
@pytest.mark.parametrize('path', BATCHED, ids=[os.path.basename(c)[4:-4] for c in BATCHED])
def test_batched_update_matches_reference_replicas_on_gpu(path, saved):
    """The E > 1 update at real shapes (n_step 60 on the line, 120 on the 5x5 grid) through the REAL kernels: E = 4
    lock-stepped replicas == mean of 4 independent reference models' gradients -> clip -> one RMSProp step
    (policies.py:20-48, 232-273; agents/utils.py:763-775, 837-855; tests/golden/make_golden_nn.py run_batched).
    Forward rtol 1e-4, post-update weights rtol 1e-3.  saved_acts: the rollout's step kernel hands gates /
    states / LSTM inputs to the update (no forward pass there) -- what BatchedTrainer does for uncoupled nets."""
    from helpers import build_product_batched, compare_batched, drive_batched
    z = load_npz(path)
    model = build_product_batched(z, 'cuda')
    if saved and not model.policy.can_save_acts:
        pytest.skip('coupled net: the update recomputes its forward pass')
    np.testing.assert_allclose(var_stats_from_named(model.policy.params.ref_variables()), z['stats0'], rtol=1e-6, atol=1e-7)
    compare_batched(drive_batched(model, z, saved=saved), z)


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ia2c', 'ma2c_cu'])
def test_batched_update_on_compact_observations_matches_reference_replicas_on_gpu(agent, monkeypatch):
    """BatchedTrainer's exact configuration on CACC -- compact observations + saved activations, the policy + value launch
    running the input encoders itself (IA2C-FP: both, lstm_step_x_kernel<3,0,1>; round 6: IA2C and ConseNet their one, <3,0,2>;
    nmarl_lstm_step_x_enc: no encoder launch exists) -- against the K = 4 reference-replica golden, and the in-kernel encoders
    against the separate ones."""
    from helpers import build_product_batched, compare_batched, drive_batched
    z = load_npz(os.path.join(GOLDEN, 'nnb_%s_line.npz' % agent))
    model = build_product_batched(z, 'cuda')
    out = drive_batched(model, z, saved=True, compact=True)
    assert model.policy.enc_in_kernel(model.E, True)
    compare_batched(out, z)
    monkeypatch.setenv('NMARL_INKERNEL_ENCODE', '0')
    model2 = build_product_batched(z, 'cuda')
    out2 = drive_batched(model2, z, saved=True, compact=True)
    assert not model2.policy.enc_in_kernel(model2.E, True)
    for k in out:                                    # matrix-core summation order vs the fmaf chain of the fc kernels
        np.testing.assert_allclose(out[k], out2[k], rtol=2e-5, atol=2e-6, err_msg=k)
