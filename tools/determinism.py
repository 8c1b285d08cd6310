"""Run-to-run determinism over a longer horizon: the same agent trained twice for B batches in one process (hipGraph rollout and
update); prints the first batch at which the flat parameter vectors differ.
    python tools/determinism.py [agent] [scenario] [batches] [E]
scenario: catchup | slowdown (CACC, n_step 60) | grid (BASELINE configs[3]: 5 x 5 synthetic ATSC grid, n_step 120)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from helpers import cacc_config, grid_config
from deeprl_network_amd.envs import make_batch_env
from deeprl_network_amd.main import init_agent
from deeprl_network_amd.utils import BatchedTrainer, Counter

agent = sys.argv[1] if len(sys.argv) > 1 else 'ma2c_nc'
scenario = sys.argv[2] if len(sys.argv) > 2 else 'slowdown'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 300
E = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
every = 10
runs = []
for k in range(2):
    if scenario == 'grid':
        cp = grid_config(agent=agent, seed=12)
    else:
        cp = cacc_config(agent=agent, scenario=scenario, seed=12, n_step=60, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=E)
    np.random.seed(12)
    model = init_agent(env, cp['MODEL_CONFIG'], 10 ** 9, 12, num_envs=E)
    tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
    snaps = []
    for b in range(1, B + 1):
        tr.run_batch()
        if b % every == 0 or b <= 5:
            snaps.append((b, model.policy.params.flat.clone()))
    torch.cuda.synchronize()
    runs.append(snaps)
    print('run %d: hand-off fallbacks %s, update captured %s' % (k, getattr(tr, 'handoff_fallbacks', None), tr._upd is not None))
    del env, model, tr
first = None
for (b, a), (_, c) in zip(*runs):
    if not torch.equal(a, c):
        first = (b, float((a - c).abs().max()), int((a != c).sum()))
        break
print('%s %s E=%d: %s' % (agent, scenario, E, 'identical over %d batches' % B if first is None else 'first difference at batch %d (max |d| %.3g, %d parameters)' % first))
