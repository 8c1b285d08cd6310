"""The synthetic-grid step kernel alone (nmarl_grid_step, compact observation, auto-reset) at E = 1024 and 2^17: us per launch and
the fraction of the 8 TB/s peak on the 3 708-byte-per-replica-step formula of bench.py."""
import os
import sys
import configparser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeprl_network_amd.envs import make_batch_env

cp = configparser.ConfigParser()
cp.read(os.path.join(ROOT, 'config', 'config_ma2c_cnet_grid.ini'))
for E in (1024, 1 << 15, 1 << 17):
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=E)
    if hasattr(env, 'set_compact_obs'):
        env.set_compact_obs(True)
    env.reset()
    e = torch.arange(E, device='cuda')[:, None]
    a = torch.arange(env.n_agent, device='cuda')[None, :]
    acts = [((e + 3 * a + s) % 5).to(torch.uint8).contiguous() for s in range(4)]
    for s in range(10):
        env.step(acts[s % 4], auto_reset=True)
    torch.cuda.synchronize()
    n = 200 if E <= (1 << 15) else 60
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for s in range(n):
        env.step(acts[s % 4], auto_reset=True)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / n
    print('E = %7d: %8.2f us per step = %.3f of 8 TB/s on 3708 B per replica-step' % (E, us, E * 3708 / us / 8e6))
    del env
