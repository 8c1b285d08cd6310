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
