"""Pin oracle/nn_ref.py (per-agent CPU restatement of the reference's networks / loss / optimiser /
buffers) against the golden vectors produced by the REAL reference model code on the fake-TF shim."""
import glob
import os
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN, cacc_config, compare_scripted, drive_scripted, load_npz, var_stats_from_named

# the heterogeneous (`*_ragged`) goldens are compared with the PRODUCT directly (tests/test_models_cpu.py,
# tests/test_gpu_models.py): they are outputs of the reference's own identical=False code, and oracle/nn_ref.py
# restates the identical-agent nets only
CASES = [c for c in sorted(glob.glob(os.path.join(GOLDEN, 'nn_*.npz'))) if not c.endswith('_ragged.npz')]


class Adapter:
    """Gives an oracle model the few attributes helpers.drive_scripted reads from the product model."""

    def __init__(self, ref, per_agent):
        self.ref, self.per_agent_optimizer, self.n_a = ref, per_agent, ref.A
        self.policy = types.SimpleNamespace(params=types.SimpleNamespace(ref_variables=self._vars))

    def _vars(self):
        return [(k, p.detach().numpy()) for k, p in self.ref.vars.v.items()]

    def reset(self):
        self.ref.reset()

    def forward(self, *a, **k):
        return self.ref.forward(*a, **k)

    def add_transition(self, *a):
        self.ref.add_transition(*a)

    def backward(self, R, dt):
        self.ref.backward(R, dt)
        loss = torch.tensor([l for l, _ in self.ref.last])
        gn = torch.tensor([g for _, g in self.ref.last])
        self.last_loss = (None, None, None, loss)
        self.grad_norm = gn if self.per_agent_optimizer else gn.repeat(self.ref.N)
        sf = self.ref.states_fw
        sf = torch.stack(sf) if isinstance(sf, list) else sf
        H = self.ref.H
        self.c_fw, self.h_fw = sf[:, None, :H], sf[:, None, H:]


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(c)[3:-4] for c in CASES])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_oracle_nn_matches_reference_graph(path, dtype):
    from oracle.nn_ref import REF_MODELS
    z = load_npz(path)
    agent, topo = str(z['agent']), str(z['topo'])
    cp = cacc_config(agent=agent, n_step=int(z['n_step']), reward_norm=float(z['reward_norm']),
                     coop_gamma=float(z['coop_gamma']))
    nb, dist = z['nb'], z['dist']
    N = nb.shape[0]
    n_feat, A = (5, 4) if topo == 'line' else (12, 5)
    is_ma = agent.startswith('ma2c')
    n_s_ls = [n_feat if is_ma else n_feat * (1 + int(nb[i].sum())) for i in range(N)]
    np.random.seed(int(z['seed']))
    ref = REF_MODELS[agent](n_s_ls, [A] * N, nb, dist, float(z['coop_gamma']), cp['MODEL_CONFIG'], dtype=dtype)
    model = Adapter(ref, per_agent=not is_ma)
    named = model.policy.params.ref_variables()
    assert [n for n, _ in named] == [str(n) for n in z['names']]
    np.testing.assert_allclose(var_stats_from_named(named), z['stats0'], rtol=1e-6, atol=1e-7)
    out = drive_scripted(model, z)
    if dtype == torch.float64:   # same precision as the shim run: agreement to rounding
        compare_scripted(out, z, rtol_fw=1e-9, rtol_w=1e-7)
    else:
        compare_scripted(out, z)
