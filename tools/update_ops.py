"""Which aten (library / elementwise) launches does one A2C update still make?  torch.profiler over `model.update()` of a warmed-up
BatchedTrainer, grouped by op and by the innermost frame of this repo that issued it.
    python tools/update_ops.py [config.ini]          -> stdout"""
import collections
import configparser
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from deeprl_network_amd.envs import make_batch_env  # noqa: E402
from deeprl_network_amd.main import AGENTS  # noqa: E402
from deeprl_network_amd.utils import BatchedTrainer, Counter  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'config', 'config_ia2c_fp_catchup.ini')
cp = configparser.ConfigParser()
cp.read(cfg)
E = cp.getint('ENV_CONFIG', 'num_envs', fallback=4096)
env = make_batch_env(cp['ENV_CONFIG'], num_envs=E, device='cuda')
np.random.seed(1)
model = AGENTS[env.agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9, cp['MODEL_CONFIG'],
                          seed=1, num_envs=E)
tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
for _ in range(3):
    tr.run_batch()
tr.rollout()
model.load_rewards(tr.buf_rraw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.update(tr.R_end, rotate=False)
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != 'CPU' or not ev.name.startswith('aten::'):
        continue
    kern = [k for k in ev.kernels] if hasattr(ev, 'kernels') else []
    if not kern:
        continue
    where = next((f for f in ev.stack if 'deeprl_network_amd' in f), ev.stack[0] if ev.stack else '?')
    where = where.replace(ROOT + '/', '')
    r = rows[(ev.name, where)]
    r[0] += len(kern)
    r[1] += sum(k.duration for k in kern)
print('%-28s %5s %9s  %s' % ('op', 'n', 'us', 'issued from'))
for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print('%-28s %5d %9.1f  %s' % (name, n, us, where[:110]))
