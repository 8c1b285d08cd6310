"""Per-batch durations right behind the warm-up (does the timed region of bench.py start at steady state?): the default job,
W warm-up batches, a device synchronisation, then B batches each between two HIP events on the launch stream.
    python tools/batch_times.py [W=3] [B=40]"""
import os
import sys
import configparser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from deeprl_network_amd.envs import make_batch_env
from deeprl_network_amd.main import AGENTS
from deeprl_network_amd.utils import BatchedTrainer, Counter

W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cp = configparser.ConfigParser()
cp.read(os.path.join(ROOT, 'config', 'config_ia2c_fp_catchup.ini'))
env = make_batch_env(cp['ENV_CONFIG'], num_envs=4096)
np.random.seed(12)
model = AGENTS[env.agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9, cp['MODEL_CONFIG'], seed=12, num_envs=4096)
tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
for _ in range(W):
    tr.run_batch()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(B + 1)]
ev[0].record()
for b in range(B):
    tr.run_batch()
    ev[b + 1].record()
torch.cuda.synchronize()
ms = [ev[b].elapsed_time(ev[b + 1]) for b in range(B)]
print('W = %d; batch durations (ms): %s' % (W, ' '.join('%.2f' % x for x in ms)))
print('mean of the first 20: %.3f ms, of batches 21..%d: %.3f ms' % (sum(ms[:20]) / 20, B, sum(ms[20:]) / max(1, B - 20)))

# the two graphs of a batch as they run IN the alternating loop (an event between them) and each replayed on its own
if tr.graph is not None and tr._upd is not None and tr._upd['epilogue_inside'] and tr._upd['apply'] is None:
    E3 = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(B)]
    for b in range(B):
        E3[b][0].record()
        tr.rollout()
        E3[b][1].record()
        tr._update()
        E3[b][2].record()
        tr.n_batches += 1
    torch.cuda.synchronize()
    r_in = sum(e[0].elapsed_time(e[1]) for e in E3[5:]) / (B - 5)
    u_in = sum(e[1].elapsed_time(e[2]) for e in E3[5:]) / (B - 5)

    def alone(fn, n=20):
        fn(); torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b_.record(); torch.cuda.synchronize()
        return a.elapsed_time(b_) / n
    snap = tr._snapshot()
    r_al = alone(tr.graph.replay)
    u_al = alone(tr._upd['grads'].replay)
    tr._restore(snap)
    print('in the alternating loop: rollout graph %.3f ms + update graph %.3f ms = %.3f ms per batch' % (r_in, u_in, r_in + u_in))
    print('each replayed on its own (20 x back to back): rollout graph %.3f ms, update graph %.3f ms = %.3f ms' % (r_al, u_al, r_al + u_al))
