"""The reference's FULL training schedule (TRAIN_CONFIG total_step = 1e6 environment steps, utils.py:213-254) at E = 1
through the CPU restatement of the reference loop (oracle/trainer_ref.py + oracle/nn_ref.py + oracle/cacc_ref.py; TensorFlow
1.12 cannot run here), INCLUDING the deterministic test episode the reference runs after every CACC training episode
(utils.py:246-251: argmax policy, train_mode off, seed of the episode just trained on).  One JSON line per training episode:
the row the reference appends to train_reward.csv plus collision flags.  Test infrastructure (imports oracle/): its output
is the E = 1 column of profiles/r03_nc_quality.md, against which the batched product's runs at E = 8 ... 4096 are read.

    python tests/learning/port_full_schedule.py ma2c_nc slowdown 12 1000000 out.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

torch.set_num_threads(1)
from helpers import cacc_config  # noqa: E402
from oracle import trainer_ref  # noqa: E402


def perform(tr):
    """Trainer.perform(-1) (utils.py:195-211) for CACC: greedy actions, policy forward only."""
    env, model = tr.env, tr.model
    ob = env.reset()
    model.reset()
    done = True
    rewards = []
    while True:
        policy, action = tr._get_policy(ob, done, mode='test')
        env.update_fingerprint(policy)
        ob, _, done, g = env.step(action)
        rewards.append(g)
        if done:
            break
    return float(np.mean(rewards)), float(np.std(rewards)), len(rewards)


def main():
    agent, scenario, seed, total, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(float(sys.argv[4])), sys.argv[5]
    norm = 800.0 if agent.startswith('ia2c') else 5000.0
    cp = cacc_config(agent=agent, scenario=scenario, seed=seed, n_step=60, reward_norm=norm, total_step=total)
    env, model, tr = trainer_ref.build(cp)
    T = env.T
    step = 0
    t0 = time.time()
    with open(out, 'a') as f:
        while step < total:
            env.train_mode = True
            ob = env.reset()
            model.reset()
            done = True
            n0 = len(tr.log)
            while True:
                ob, done, R = tr.explore(ob, done)
                model.backward(R, 0)
                if done:
                    break
            ep = tr.log[n0:]
            del tr.log[:]
            step += len(ep)
            g = np.array([x[1] for x in ep])
            env.train_mode = False
            m, s, n = perform(tr)
            env.train_mode = True
            row = dict(step=step, train_avg_reward=float(g.mean()), train_len=len(ep), train_collision=int(len(ep) < T),
                       avg_reward=m, std_reward=s, test_len=n, test_collision=int(n < T), wall_s=round(time.time() - t0, 1))
            f.write(json.dumps(row) + '\n')
            f.flush()


if __name__ == '__main__':
    main()
