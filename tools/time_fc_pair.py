"""The update's encoder backward at the BASELINE size (8 x 245 760 rows): two nmarl_fc_bwd_gather launches vs nmarl_fc_bwd_pair reading S vs reading the sign image."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeprl_network_amd import ops
N, rows = 8, 60 * 4096
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g).cuda()
nbr_idx = -torch.ones(N, 2, dtype=torch.int32)
for i in range(N):
    l = [j for j in (i - 1, i + 1) if 0 <= j < N]
    nbr_idx[i, :len(l)] = torch.tensor(l, dtype=torch.int32)
nbr_self = torch.cat([torch.arange(N, dtype=torch.int32).view(-1, 1), nbr_idx], dim=1).cuda()
xs = [r(rows, N, 5).transpose(0, 1), torch.softmax(r(N, rows, 4), dim=-1)]
idxs = [nbr_self, nbr_idx.cuda()]
ws = [r(N, 15, 64) * 0.3, r(N, 8, 64) * 0.3]; bs = [r(N, 64) * 0.1, r(N, 64) * 0.1]
S = ops.fc_fwd_multi([(x, w, b, i) for x, w, b, i in zip(xs, ws, bs, idxs)], ops.BIAS_RELU)
dS = r(N, rows, 128)
bits = ops.relu_bits_pack(S)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
two = lambda: [ops.fc_bwd(x, S[:, :, 64 * k:64 * k + 64], dS[:, :, 64 * k:64 * k + 64], ops.BIAS_RELU, nbr_idx=i) for k, (x, i) in enumerate(zip(xs, idxs))]
print('two launches + two reduces   %.1f us' % t(two))
print('pair (S)                      %.1f us' % t(lambda: ops.fc_bwd_pair(xs, idxs, S, dS, ops.BIAS_RELU)))
print('pair (bits)                   %.1f us  = %.2f TB/s on dS + bits' % ((lambda u: (u, (dS.numel() * 4 + bits.numel() * 4) / u / 1e6))(t(lambda: ops.fc_bwd_pair(xs, idxs, None, dS, ops.BIAS_RELU, bits=bits)))))
