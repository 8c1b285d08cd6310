"""The drop-in E = 1 path on the MI355X: the reference's own config (no num_envs) through `Trainer` -- one replica, the
reference's call order (forward 'p', np.random draws, forward 'v', env.step, add_transition; backward per n_step), every step a
handful of small launches and host <-> device round trips.  Timed like bench.py's cpu_baseline times the CPU port
(oracle/trainer_ref.py run_batches: explore + backward cycles, new episodes as needed, no test episodes), so the two
env-steps/s figures are the same quantity.  Also: C-ABI calls per env step (counted at the binding).
    python tools/e1_path.py [config.ini] [batches=100]
Under `rocprofv3 --kernel-trace --stats` the kernel count of the run / the printed env steps = launches per env step."""
import collections
import configparser
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deeprl_network_amd import _lib  # noqa: E402
from deeprl_network_amd.envs import init_env  # noqa: E402
from deeprl_network_amd.main import init_agent  # noqa: E402
from deeprl_network_amd.utils import Counter, Trainer  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'config', 'config_ia2c_catchup.ini')
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cp = configparser.ConfigParser()
cp.read(cfg)
env = init_env(cp['ENV_CONFIG'])
model = init_agent(env, cp['MODEL_CONFIG'], int(1e9), cp.getint('ENV_CONFIG', 'seed'))
tr = Trainer(env, model, Counter(int(1e12), int(1e12), int(1e12)), None)

calls = collections.Counter()
for name in _lib.SIGNATURES:
    fn = getattr(_lib.lib, name)

    def counted(*a, _fn=fn, _name=name):
        calls[_name] += 1
        return _fn(*a)
    setattr(_lib.lib, name, counted)


def run(n):
    """n explore + backward cycles (the CPU port's run_batches)."""
    steps, done, ob = 0, True, None
    for _ in range(n):
        if done:
            ob = env.reset()
            model.reset()
            tr.cur_step, tr.episode_rewards = 0, []
        s0 = tr.cur_step
        ob, done, R = tr.explore(ob, done)
        model.backward(R, env.T - tr.cur_step)
        steps += tr.cur_step - s0
    return steps


run(2)                                    # warm-up: allocator, library handles, lazily built tables
torch.cuda.synchronize()
calls.clear()
t0 = time.perf_counter()
steps = run(batches)
torch.cuda.synchronize()
sec = time.perf_counter() - t0
n_abi = sum(v for k, v in calls.items() if not k.endswith(('_floats', '_words', '_chunks', '_parts', '_blocks', '_version', '_capacity')))
print('E = 1 drop-in path, %s (%s, %d agents): %d n_step batches = %d env steps in %.2f s -> %.0f env-steps/s (agents x steps/s), '
      '%.2f ms per env step, %.1f updates/s; %.1f C-ABI launches per env step'
      % (os.path.basename(cfg), env.agent, env.n_agent, batches, steps, sec, steps * env.n_agent / sec, sec / steps * 1e3, batches / sec,
         n_abi / steps))
print('E1PATH steps %d seconds %.6f agents %d' % (steps, sec, env.n_agent))
top = ', '.join('%s %.1f' % (k[6:], v / steps) for k, v in calls.most_common(8))
print('  calls per env step: ' + top)
