"""Kernel-only microbenchmark of the CACC step kernel (actions = (env+3*agent+step) mod 4,
SURVEY.md 8d).  Prints achieved algorithmic GB/s (B_alg = 41*N+19 = 347 B per replica-step)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import cacc_config
from deeprl_network_amd.envs.cacc_env import CACCBatchEnv

B_ALG = 41 * 8 + 19 + 4        # compact observation + the [E] reward vector = 351 B (bench.py B_ALG_COMPACT)

for E in [4096, 32768, 1 << 18, 1 << 20, 1 << 21, 1 << 22]:
    env = CACCBatchEnv(cacc_config()['ENV_CONFIG'], num_envs=E)
    env.set_compact_obs(True)
    env.reset()
    e = torch.arange(E, device='cuda')[:, None]
    a = torch.arange(8, device='cuda')[None, :]
    acts = [((e + 3 * a + s) % 4).to(torch.uint8).contiguous() for s in range(4)]
    for s in range(20):
        env.step(acts[s % 4], auto_reset=True)
    torch.cuda.synchronize()
    n = 200 if E <= (1 << 20) else 50
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for s in range(n):
        env.step(acts[s % 4], auto_reset=True)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / n
    print('E=%8d  %.2f us/launch  %.1f GB/s algorithmic  (%.2f%% of 8 TB/s)  %.1f M agent-steps/s' %
          (E, us, B_ALG * E / us / 1e3, B_ALG * E / us / 1e3 / 80, 8 * E / us))
