"""Does configs[3] learn?  One full-schedule run of config/config_ma2c_cnet_grid.ini (CommNet on the synthetic 5 x 5 grid, 25 agents x
1024 replicas, total_step 1e6 lock-steps = 8 333 updates) through BatchedTrainer, against fixed controllers ON THE SAME ENV:
  greedy   the reference's LargeGridController.greedy (envs/large_grid_env.py:41-45: the phase serving the most waiting vehicles)
           restated on the 12-link observation: score(phase) = sum of the wave of the links the phase serves (G or g in
           large_grid_env.py:25-26), arg max -- the reference hard-codes the same sums for its 6-lane observation
  random   uniform phases;   fixed0 always phase 0;   cyclic phase = step mod 5
Metric: mean per-step global reward of an episode (= - total queue over the 25 nodes, atsc_env.py:383-418), the number
BatchedTrainer.stats() logs.      python tools/learn_grid.py [batches] [log every]   -> gpurun_out/r04_learn_grid.json"""
import configparser
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from deeprl_network_amd.envs import make_batch_env  # noqa: E402
from deeprl_network_amd.main import AGENTS  # noqa: E402
from deeprl_network_amd.utils import BatchedTrainer, Counter  # noqa: E402

PHASES = ['GGgrrrGGgrrr', 'rrrGrGrrrGrG', 'rrrGGrrrrGGr', 'rrrGGGrrrrrr', 'rrrrrrrrrGGG']      # large_grid_env.py:25-26

cp = configparser.ConfigParser()
cp.read(os.path.join(ROOT, 'config', 'config_ma2c_cnet_grid.ini'))
n_step = cp.getint('MODEL_CONFIG', 'batch_size')
total = int(cp.getfloat('TRAIN_CONFIG', 'total_step'))
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else total // n_step
every = int(sys.argv[2]) if len(sys.argv) > 2 else 250
E = cp.getint('ENV_CONFIG', 'num_envs')
seed = cp.getint('ENV_CONFIG', 'seed')


def controller_episode(kind, n_envs=256):
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=n_envs, device='cuda', seed=seed + 1, env_id_base=10 ** 8)
    env.reset()
    serve = torch.tensor([[0.0 if ch == 'r' else 1.0 for ch in p] for p in PHASES], device='cuda')          # [5,12]
    tot = torch.zeros(n_envs, dtype=torch.float64, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(1)
    for t in range(env.T):
        own = env.obs[:, :, :12]                                  # a node's own 12 link waves lead its observation
        if kind == 'greedy':
            a = torch.einsum('enk,pk->enp', own, serve).argmax(dim=-1)
        elif kind == 'random':
            a = torch.randint(0, 5, (n_envs, 25), device='cuda', generator=g)
        elif kind == 'cyclic':
            a = torch.full((n_envs, 25), t % 5, device='cuda')
        else:
            a = torch.zeros(n_envs, 25, dtype=torch.long, device='cuda')
        _, _, d, gr = env.step(a.to(torch.uint8).contiguous())
        tot += gr.double()
    return float((tot / env.T).mean().item()), float((tot / env.T).std().item())


out = {'config': 'config_ma2c_cnet_grid.ini', 'E': E, 'n_step': n_step, 'batches': n_batches, 'controllers': {}}
for kind in ('greedy', 'random', 'cyclic', 'fixed0'):
    m, s = controller_episode(kind)
    out['controllers'][kind] = {'avg_reward': m, 'std_over_replicas': s}
    print('controller %-7s: mean per-step global reward %.2f (std over replicas %.2f)' % (kind, m, s), flush=True)

env = make_batch_env(cp['ENV_CONFIG'], num_envs=E, device='cuda')
np.random.seed(seed)
model = AGENTS[env.agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, total,
                          cp['MODEL_CONFIG'], seed=seed, num_envs=E)
tr = BatchedTrainer(env, model, Counter(total, 10 ** 12, 10 ** 12), use_graph=True)
rows = []
t0 = time.time()
for b in range(1, n_batches + 1):
    tr.run_batch()
    if b % every == 0 or b == n_batches:
        st = tr.stats()
        rows.append(dict(batch=b, lock_steps=b * n_step, env_steps=b * n_step * E * env.n_agent, episodes=st['episodes'],
                         train_avg_reward=st['avg_reward'], train_std_reward=st['std_reward'], wall_s=round(time.time() - t0, 1)))
        print(json.dumps(rows[-1]), flush=True)
out['rows'] = rows
out['handoff_fallbacks'] = tr.handoff_fallbacks
# the trained policy, deterministic (argmax) and stochastic, on fresh replicas
m, s, _ = tr.evaluate(n_envs=256)
out['trained_argmax'] = {'avg_reward': m, 'std_over_replicas': s, 'action_share': tr.last_eval_action_share}
print('trained policy, argmax test episodes: %.2f (std %.2f), action share %s' % (m, s, tr.last_eval_action_share))
path = os.path.join(ROOT, 'gpurun_out', 'r04_learn_grid.json')
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, 'w'), indent=1)
print('written', path)
