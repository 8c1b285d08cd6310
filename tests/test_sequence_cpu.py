"""The manual-BPTT coupled sequences (agents/sequence.py: NeurComm, CommNet, DIAL) against the plain per-step
autograd unroll of the same policy, on CPU with emulated ops: outputs and every parameter gradient must agree."""
import numpy as np
import pytest
import torch

from cpu_emulation import cpu_ops


def _masks(kind):
    if kind == 'line':
        idx = np.arange(8)
        return (np.abs(idx[:, None] - idx[None, :]) == 1).astype(int), 5, 4
    d = np.array([[abs(i // 5 - j // 5) + abs(i % 5 - j % 5) for j in range(25)] for i in range(25)])
    return (d == 1).astype(int), 12, 5


@pytest.mark.parametrize('cls_name', ['NCMultiAgentPolicy', 'IC3MultiAgentPolicy', 'DIALMultiAgentPolicy'])
@pytest.mark.parametrize('topo', ['line', 'grid'])
@pytest.mark.parametrize('masked', [None, (0,)])
def test_manual_bptt_equals_autograd(cls_name, topo, masked):
    from deeprl_network_amd.agents import policies
    nb, n_feat, A = _masks(topo)
    T, E = 4, 3
    with cpu_ops():
        np.random.seed(5)
        pol = getattr(policies, cls_name)(n_feat, A, nb, device='cpu')
        pol.params.init_reference_order()
        g = torch.Generator().manual_seed(1)
        N = pol.N
        X = torch.randn(T, E, N, pol.n_obs, generator=g) * 0.5
        FP = torch.softmax(torch.randn(N, T * E, A, generator=g), -1)
        done = torch.zeros(T, E)
        done[0, 1] = 1.0
        if masked is None:
            done[2, 0] = 1.0                      # a mid-batch episode start (reference API semantics)
        h0, c0 = torch.randn(N, E, 64, generator=g) * 0.3, torch.randn(N, E, 64, generator=g) * 0.3
        w = torch.randn(N, T * E, 64, generator=g)
        grads = []
        outs = []
        for fused in (True, False):
            pol.fused_coupled = fused
            pol.params.grad.zero_()
            hh, cc = h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
            Hs = pol.unroll(X, FP, done, hh, cc, masked_steps=masked)
            (Hs * w).sum().backward()
            outs.append(Hs.detach().clone())
            grads.append((pol.params.grad.clone(), hh.grad.clone(), cc.grad.clone()))
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-5, atol=1e-6)
    for a, b, name in zip(grads[0], grads[1], ['params', 'h0', 'c0']):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5, msg=name)
    assert grads[0][0].abs().sum() > 0
