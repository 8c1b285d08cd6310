"""CPU baseline of the ENV-ONLY hot path with the REAL reference: the imported,
unmodified /root/reference/envs/cacc_env.py `CACCEnv.step` (cacc_env.py:191-242) driven by pre-generated actions,
  * one process pinned to one core,
  * one process per host core (independent replicas, no communication),
for the IA2C-FP and MA2C observation forms -> agent-steps/s (= agents x env steps / s).

    python tools/cpu_env_baseline.py [--seconds 10] [--out profiles/r04_cpu_env_reference.json]

/root/reference exists only in the authoring container (never on the GPU box), so the result is committed under
profiles/ and bench.py reports it as `cpu_baseline.reference_env_only` next to the restated full loop it times live."""
import argparse
import configparser
import io
import json
import multiprocessing as mp
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'


def _config(agent, scenario):
    from helpers import CACC_INI
    cp = configparser.ConfigParser()
    cp.read_file(io.StringIO(CACC_INI.format(agent=agent, scenario=scenario, seed=12, coop_gamma=-1, n_step=60,
                                             reward_norm=800.0, total_step=1200)))
    return cp['ENV_CONFIG']


def _worker(core, agent, scenario, seconds, q):
    if core is not None and hasattr(os, 'sched_setaffinity'):
        os.sched_setaffinity(0, {core})
    sys.path.insert(0, REF)
    import logging
    logging.disable(logging.CRITICAL)
    from envs.cacc_env import CACCEnv                       # the reference, unmodified
    env = CACCEnv(_config(agent, scenario))
    env.train_mode = True
    rng = np.random.RandomState(7 + (core or 0))
    tape = rng.randint(0, 4, size=(env.T, env.n_agent))      # pre-generated actions (no policy in the timed loop)
    env.reset()
    for k in range(20):                                      # warm-up
        env.step(tape[k])
    env.reset()
    steps, t, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, _, done, _ = env.step(tape[t])
        steps += 1
        t += 1
        if done:
            env.reset()
            t = 0
    q.put((steps, time.perf_counter() - t0, env.n_agent))


def run(agent, scenario, cores, seconds):
    q = mp.Queue()
    ps = [mp.Process(target=_worker, args=(c, agent, scenario, seconds, q)) for c in cores]
    for p in ps:
        p.start()
    res = [q.get() for _ in ps]
    for p in ps:
        p.join()
    return sum(s * n / sec for s, sec, n in res), sum(s for s, _, _ in res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=10.0)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r02_cpu_env_reference.json'))
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, 'envs')):
        raise SystemExit('needs the reference checkout at %s' % REF)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(ncores))
    out = {'what': 'reference envs/cacc_env.py CACCEnv.step (cacc_env.py:191-242), imported unmodified, pre-generated actions',
           'host': platform.processor() or platform.machine(), 'cpu_model': _cpu_model(), 'cores_available': ncores,
           'numpy': np.__version__, 'seconds_per_run': args.seconds, 'unit': 'agent-steps/s (agents x env steps / s)', 'runs': {}}
    for agent, scenario in (('ia2c_fp', 'catchup'), ('ma2c_nc', 'slowdown')):
        one, n1 = run(agent, scenario, cores[-1:], args.seconds)
        allc, na = run(agent, scenario, cores, args.seconds)
        out['runs']['%s_%s' % (agent, scenario)] = {'one_core': one, 'one_core_env_steps': n1, 'all_cores': allc,
                                                    'all_cores_processes': len(cores), 'all_cores_env_steps': na}
        print('%-18s 1 core: %8.0f agent-steps/s   %d cores: %8.0f agent-steps/s' % (agent + '/' + scenario, one, len(cores), allc))
    json.dump(out, open(args.out, 'w'), indent=1)
    print('wrote', args.out)


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


if __name__ == '__main__':
    main()
