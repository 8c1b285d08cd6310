"""Feasibility: do two half-size fused LSTM steps on two streams of one hipGraph overlap their load / MFMA / store
phases (vs one full-size launch)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeprl_network_amd import ops

N, H = 8, 64


def mk(E):
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    return dict(h=r(N, E, H) * .3, c=r(N, E, H) * .3, z=r(N, E, 4 * H), done=torch.zeros(E, device='cuda'))


wh = (torch.randn(N, H, 4 * H) * 0.1).cuda(); b = torch.zeros(N, 4 * H).cuda()


def step(d):
    ops.lstm_step_fused(d['h'], wh, b, d['z'], None, d['c'], d['done'], None, d['c'], d['h'])


def run(shards, reps=20):
    data = [mk(4096 // shards) for _ in range(shards)]
    streams = [torch.cuda.Stream() for _ in range(shards)]
    for d in data: step(d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for s, d in zip(streams, data):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(reps): step(d)
        for s in streams: cur.wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * reps)


for sh in (1, 2, 4):
    print('%d shard(s) of E=%d on %d stream(s): %.2f us per lock-step-equivalent (all shards)' % (sh, 4096 // sh, sh, run(sh)))
