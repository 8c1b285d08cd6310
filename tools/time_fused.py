"""Time nmarl_lstm_step_fused at the bench shape (N=8, E=4096, H=64); NMARL_FUSED_VARIANT=1|2 picks the kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeprl_network_amd import ops
N, E, H = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64
g = torch.Generator().manual_seed(0)
h = torch.randn(N, E, H, generator=g).cuda(); c = torch.randn(N, E, H, generator=g).cuda()
z1 = torch.randn(N, E, 4 * H, generator=g).cuda(); wh = (torch.randn(N, H, 4 * H, generator=g) * 0.2).cuda()
b = torch.zeros(N, 4 * H).cuda(); done = torch.zeros(E).cuda()
co, ho = torch.empty_like(c), torch.empty_like(h)
gates = torch.empty(N, E, 4 * H, device='cuda')
for with_gates in (False, True):
    f = lambda: ops.lstm_step_fused(h, wh, b, z1, None, c, done, gates if with_gates else None, co, ho)
    for _ in range(5): f()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): f()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): gr.replay()
    t1.record(); torch.cuda.synchronize()
    print('variant %s gates=%s: %.2f us per call' % (os.environ.get('NMARL_FUSED_VARIANT', '2'), with_gates, t0.elapsed_time(t1) * 1e3 / 200))
