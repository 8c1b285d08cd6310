import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_sequence_cpu import _masks
from cpu_emulation import cpu_ops
from deeprl_network_amd.agents import policies
topo = sys.argv[2] if len(sys.argv) > 2 else 'grid'
nb, n_feat, A = _masks(topo)
T, E = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(1)
def build(dev):
    np.random.seed(5)
    pol = getattr(policies, sys.argv[3] if len(sys.argv) > 3 else 'NCMultiAgentPolicy')(n_feat, A, nb, device=dev)
    pol.params.init_reference_order()
    return pol
pol = build('cuda'); N = pol.N
X = torch.randn(T, E, N, pol.n_obs, generator=g) * 0.5
FP = torch.softmax(torch.randn(N, T * E, A, generator=g), -1)
done = torch.zeros(T, E); done[0, ::3] = 1.0
h0, c0 = torch.randn(N, E, 64, generator=g) * 0.3, torch.randn(N, E, 64, generator=g) * 0.3
w = torch.randn(N, T * E, 64, generator=g)
def run(pol, dev, fused):
    pol.fused_coupled = fused
    pol.params.grad.zero_()
    hh, cc = h0.to(dev).clone().requires_grad_(True), c0.to(dev).clone().requires_grad_(True)
    Hs = pol.unroll(X.to(dev), FP.to(dev), done.to(dev), hh, cc, masked_steps=(0,))
    (Hs * w.to(dev)).sum().backward()
    return pol.params.grad.detach().cpu().clone()
gm = run(pol, 'cuda', True); ga = run(pol, 'cuda', False)
with cpu_ops():
    pc = build('cpu'); gc = run(pc, 'cpu', False)
for key, (o, size, shape, fmt, _) in pol.params.index.items():
    sl = slice(o, o + size)
    print('   %-10s scale %9.3e  manual-vs-cpu %9.3e  autograd-vs-cpu %9.3e' % (key, gc[:, sl].abs().max().item(), (gm[:, sl] - gc[:, sl]).abs().max().item(), (ga[:, sl] - gc[:, sl]).abs().max().item()))
print(topo, 'E', E, 'scale %.3e manual-vs-cpu %.3e autograd-vs-cpu %.3e' % (gc.abs().max().item(), (gm - gc).abs().max().item(), (ga - gc).abs().max().item()))
ga2 = run(pol, 'cuda', False)
print('   autograd run-to-run diff %.3e' % (ga - ga2).abs().max().item())
