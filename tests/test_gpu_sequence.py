"""agents/sequence.py (manual BPTT for NeurComm / CommNet / DIAL, fused MFMA step inside) against the per-step autograd
unroll of the same policy, on the GPU through the real kernels, at a size with ragged row tiles (E = 300)."""
import numpy as np
import pytest
import torch

from test_sequence_cpu import _masks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cls_name', ['NCMultiAgentPolicy', 'IC3MultiAgentPolicy', 'DIALMultiAgentPolicy'])
@pytest.mark.parametrize('topo', ['line', 'grid'])
def test_manual_bptt_equals_autograd_gpu(cls_name, topo):
    from deeprl_network_amd.agents import policies
    nb, n_feat, A = _masks(topo)
    T, E = 5, 300
    np.random.seed(5)
    pol = getattr(policies, cls_name)(n_feat, A, nb, device='cuda')
    pol.params.init_reference_order()
    g = torch.Generator().manual_seed(1)
    N = pol.N
    X = (torch.randn(T, E, N, pol.n_obs, generator=g) * 0.5).cuda()
    FP = torch.softmax(torch.randn(N, T * E, A, generator=g), -1).cuda()
    done = torch.zeros(T, E)
    done[0, ::3] = 1.0
    done = done.cuda()
    h0, c0 = (torch.randn(N, E, 64, generator=g) * 0.3).cuda(), (torch.randn(N, E, 64, generator=g) * 0.3).cuda()
    w = torch.randn(N, T * E, 64, generator=g).cuda()
    res = []
    for fused in (True, False):
        pol.fused_coupled = fused
        pol.params.grad.zero_()
        hh, cc = h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
        Hs = pol.unroll(X, FP, done, hh, cc, masked_steps=(0,))
        (Hs * w).sum().backward()
        res.append((Hs.detach().clone(), pol.params.grad.clone(), hh.grad.clone(), cc.grad.clone()))
    for a, b, name in zip(res[0], res[1], ['Hs', 'params', 'h0', 'c0']):
        # gradients are sums over T*E = 1500 rows of O(1) terms in two different fp32 summation orders:
        # judge the error against the scale of the tensor, not element by element
        err = (a - b).abs().max().item()
        scale = b.abs().max().item()
        assert err <= 2e-5 * max(scale, 1.0), '%s: max |diff| %.3e vs scale %.3e' % (name, err, scale)
