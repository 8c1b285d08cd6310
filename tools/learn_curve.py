"""Learning sanity run: does the batched path actually learn CACC?  Trains on E replicas for a number of n_step
batches and prints training statistics (finished episodes' mean per-step global reward, collisions) plus
deterministic test episodes (train_mode off, argmax policy -- utils.py:246-251).
    python tools/learn_curve.py [agent] [scenario] [E] [batches] [log every] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from helpers import cacc_config
from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
from deeprl_network_amd.main import AGENTS
from deeprl_network_amd.utils import BatchedTrainer, Counter

agent = sys.argv[1] if len(sys.argv) > 1 else 'ia2c_fp'
scenario = sys.argv[2] if len(sys.argv) > 2 else 'catchup'
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
n_batches = int(sys.argv[4]) if len(sys.argv) > 4 else 400
every = int(sys.argv[5]) if len(sys.argv) > 5 else 50
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 12       # env seed (Philox key), weight init draws, model seed
if scenario.endswith('.ini'):                     # any shipped config: python tools/learn_curve.py ma2c_nc config/x.ini E batches
    import configparser
    from deeprl_network_amd.envs import make_batch_env
    cp = configparser.ConfigParser()
    cp.read(scenario)
    cp['ENV_CONFIG']['agent'] = agent
    cp['ENV_CONFIG']['seed'] = str(seed)
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=E)
    scenario = os.path.basename(scenario)[:-4]
else:
    cp = cacc_config(agent=agent, scenario=scenario, seed=seed, n_step=60, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
    env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
np.random.seed(seed)
model = AGENTS[agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                      cp['MODEL_CONFIG'], seed=seed, num_envs=E, n_feat_ls=getattr(env, 'n_feat_ls', None))
tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
rows = []
t0 = time.time()
m, s, c = tr.evaluate(n_envs=64)
rows.append(dict(batch=0, env_steps=0, test_avg_reward=m, test_collisions=c, test_action_share=tr.last_eval_action_share))
print(json.dumps(rows[-1]))
for b in range(1, n_batches + 1):
    tr.run_batch()
    if b % every == 0:
        st = tr.stats()
        m, s, c = tr.evaluate(n_envs=64)
        rows.append(dict(batch=b, env_steps=b * model.n_step * E * env.n_agent, lock_steps=tr.global_counter.cur_step, train_episodes=st['episodes'],
                         train_avg_reward=st['avg_reward'], train_collisions=st['collisions'],
                         test_avg_reward=m, test_std_reward=s, test_collisions=c, test_action_share=tr.last_eval_action_share,
                         wall_s=round(time.time() - t0, 1)))
        print(json.dumps(rows[-1]))
out = os.path.join(ROOT, 'gpurun_out', 'learn_%s_%s_E%d_b%d_s%d.json' % (agent, scenario, E, n_batches, seed))
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(dict(agent=agent, scenario=scenario, E=E, seed=seed, rows=rows), open(out, 'w'), indent=1)
