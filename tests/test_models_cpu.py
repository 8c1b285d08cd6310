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
