"""Is a lock-step rollout / an update's backward of a coupled net the same twice?  Every batch: the rollout graph is replayed twice from
the same state and every buffer it writes is compared (integer checksums of the bit patterns); the update's gradient is formed twice from
the same buffers and compared; then the batch is run normally.  A difference = a kernel whose result depends on timing.
    python tools/race_hunt.py [agent=ma2c_nc] [scenario=slowdown] [batches=600] [E=4096]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from helpers import cacc_config
from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
from deeprl_network_amd.main import AGENTS
from deeprl_network_amd.utils import BatchedTrainer, Counter

agent = sys.argv[1] if len(sys.argv) > 1 else 'ma2c_nc'
scenario = sys.argv[2] if len(sys.argv) > 2 else 'slowdown'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 600
E = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
cp = cacc_config(agent=agent, scenario=scenario, seed=12, n_step=60, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
np.random.seed(12)
m = AGENTS[agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9, cp['MODEL_CONFIG'], seed=12, num_envs=E)
tr = BatchedTrainer(env, m, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
for _ in range(3):
    tr.run_batch()


def bits(t):
    t = t.contiguous()
    v = t.view(torch.int32) if t.element_size() == 4 else t.to(torch.int32)
    return int(v.to(torch.int64).sum().item()), int((v.to(torch.int64) * (torch.arange(v.numel(), device=v.device).view(v.shape) % 8191 + 1)).sum().item())


def rollout_outputs():
    p = m.policy
    out = dict(act=m.buf_act, v=m.buf_v, vn=m.buf_vn, H=m.H_all, C=m.C_all, G=m.G_buf, S=m.S_buf, fp=m.buf_fp, x=m.buf_x, R_end=tr.R_end,
               rraw=tr.buf_rraw, done=m.buf_done_post, h_fw=m.h_fw, c_fw=m.c_fw)
    for k, v in getattr(p, '_extra', {}).items():
        out['extra_' + k] = v
    return out


bad_r = bad_u = 0
for b in range(B):
    snap = tr._snapshot()
    tr.rollout()
    A = {k: bits(v) for k, v in rollout_outputs().items()}
    tr._restore(snap)
    tr.rollout()
    Bv = {k: bits(v) for k, v in rollout_outputs().items()}
    diff = [k for k in A if A[k] != Bv[k]]
    if diff:
        bad_r += 1
        print('batch %d: ROLLOUT differs between two replays from the same state in %s' % (b, diff), flush=True)
    # the update's gradient twice from the same buffers (eager launches of the kernels the update graph replays)
    vn0 = m.buf_vn.clone()
    host = (m.t, getattr(m.policy, '_enc_was_saved', False), getattr(m.policy, '_mm_was_saved', False), m.policy._bits_steps)
    g = []
    for k in range(2):
        m.buf_vn.copy_(vn0)
        m.t = tr.n_step
        m.policy._enc_was_saved, m.policy._mm_was_saved, m.policy._bits_steps = host[1], host[2], host[3]
        m.load_rewards(tr.buf_rraw)
        m.update_grads(tr.R_end)
        g.append(m.policy.params.grad.clone())
    m.buf_vn.copy_(vn0)
    if not torch.equal(g[0], g[1]):
        bad_u += 1
        d = (g[0] - g[1]).abs()
        print('batch %d: GRADIENT differs between two backward passes over the same buffers: %d entries, max |d| %.3g (max |g| %.3g)'
              % (b, int((g[0] != g[1]).sum()), float(d.max()), float(g[0].abs().max())), flush=True)
    m.t, m.policy._enc_was_saved, m.policy._mm_was_saved, m.policy._bits_steps = host
    tr._restore(snap)
    tr.run_batch()
print('%s %s E=%d: %d batches, rollout differed %d times, gradient differed %d times; hand-off fallbacks %d' % (
    agent, scenario, E, B, bad_r, bad_u, tr.handoff_fallbacks))
