"""Time nmarl_net_step (hipGraph of 20 launches) at several E; NMARL_NET_REPS picks replicas per block."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from helpers import net_config
from deeprl_network_amd.envs.real_net_env import RealNetBatchEnv
for E in [int(x) for x in (sys.argv[1:] or ['1024', '8192', '65536'])]:
    env = RealNetBatchEnv(net_config()['ENV_CONFIG'], num_envs=E)
    env.reset()
    tp = env.topo
    rng = np.random.RandomState(0)
    acts = [torch.from_numpy(np.stack([rng.randint(0, tp.n_a_ls[i], size=E) for i in range(tp.N)], 1).astype(np.uint8)).cuda() for _ in range(4)]
    for k in range(40): env.step(acts[k % 4], auto_reset=True)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): env.step(acts[0], auto_reset=True)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(20): env.step(acts[k % 4], auto_reset=True)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    print('reps=%s E=%d: %.1f us per launch, %.2f TB/s algorithmic (16.6 KB/replica)' % (os.environ.get('NMARL_NET_REPS', 'auto'), E, us, 16652 * E / us / 1e6))
