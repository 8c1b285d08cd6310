"""Is the batch loop host-bound?  Host enqueue time vs device time per batch (headline config)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import configparser  # noqa: E402

from deeprl_network_amd.envs import make_batch_env  # noqa: E402
from deeprl_network_amd.main import AGENTS  # noqa: E402
from deeprl_network_amd.utils import BatchedTrainer, Counter  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'config', 'config_ia2c_fp_catchup.ini')
    cp = configparser.ConfigParser()
    cp.read(cfg)
    E = cp.getint('TRAIN_CONFIG', 'num_envs', fallback=4096)
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=E, device='cuda')
    np.random.seed(env.seed)
    model = AGENTS[env.agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 6,
                              cp['MODEL_CONFIG'], seed=env.seed, num_envs=E, device='cuda')
    tr = BatchedTrainer(env, model, Counter(10 ** 9, 10 ** 9, 10 ** 9), use_graph=True)
    for _ in range(3):
        tr.run_batch()
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    host = []
    for _ in range(K):
        a = time.perf_counter()
        tr.rollout()
        b = time.perf_counter()
        tr.model.load_rewards(tr.buf_rraw)
        tr.model.update(tr.R_end)
        c = time.perf_counter()
        host.append((b - a, c - b))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host enqueue per batch: rollout %.2f ms, update %.2f ms; loop returned after %.2f ms/batch; device done after %.2f ms/batch'
          % (1e3 * np.mean([h[0] for h in host]), 1e3 * np.mean([h[1] for h in host]), 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K))


if __name__ == '__main__':
    main()
