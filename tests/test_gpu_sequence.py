"""agents/sequence.py (manual BPTT for NeurComm / CommNet / DIAL, fused MFMA step inside) on the GPU through the real
kernels against the per-step AUTOGRAD unroll of the same policy evaluated on the CPU with the oracle op restatements
(tests/cpu_emulation.py), at a size with ragged row tiles (E = 300).

(The per-step autograd unroll on the GPU is NOT used as the reference: at N = 25, E = 300 one of the library's
batched fp32 GEMMs with K = E = 300 returns gradients that are 0.5 % off -- found with tools/dbg_seq.py, reproducible,
E = 256 is fine.  The product's fused sequences only issue weight-gradient GEMMs over all T*E rows.)"""
import numpy as np
import pytest
import torch

from cpu_emulation import cpu_ops
from test_sequence_cpu import _masks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cls_name', ['NCMultiAgentPolicy', 'IC3MultiAgentPolicy', 'DIALMultiAgentPolicy'])
@pytest.mark.parametrize('topo', ['line', 'grid'])
def test_manual_bptt_on_gpu_equals_cpu_autograd(cls_name, topo):
    from deeprl_network_amd.agents import policies
    nb, n_feat, A = _masks(topo)
    T, E = 5, 300
    g = torch.Generator().manual_seed(1)

    def build(dev):
        np.random.seed(5)
        pol = getattr(policies, cls_name)(n_feat, A, nb, device=dev)
        pol.params.init_reference_order()
        return pol
    pol = build('cuda')
    N = pol.N
    X = torch.randn(T, E, N, pol.n_obs, generator=g) * 0.5
    FP = torch.softmax(torch.randn(N, T * E, A, generator=g), -1)
    done = torch.zeros(T, E)
    done[0, ::3] = 1.0
    h0, c0 = torch.randn(N, E, 64, generator=g) * 0.3, torch.randn(N, E, 64, generator=g) * 0.3
    w = torch.randn(N, T * E, 64, generator=g)

    def run(pol, dev, fused):
        pol.fused_coupled = fused
        pol.params.grad.zero_()
        hh, cc = h0.to(dev).clone().requires_grad_(True), c0.to(dev).clone().requires_grad_(True)
        Hs = pol.unroll(X.to(dev), FP.to(dev), done.to(dev), hh, cc, masked_steps=(0,))
        (Hs * w.to(dev)).sum().backward()
        return [t.detach().cpu().clone() for t in (Hs, pol.params.grad, hh.grad, cc.grad)]
    got = run(pol, 'cuda', True)
    with cpu_ops():
        want = run(build('cpu'), 'cpu', False)
    for a, b, name in zip(got, want, ['Hs', 'params', 'h0', 'c0']):
        # gradients are sums over T*E = 1500 rows in different fp32 summation orders: judge against the tensor's scale
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert err <= 2e-5 * max(scale, 1.0), '%s: max |diff| %.3e vs scale %.3e' % (name, err, scale)
