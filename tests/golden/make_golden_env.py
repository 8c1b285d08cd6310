"""Generate tests/golden/cacc_*.npz by running the REAL reference environment
(/root/reference/envs/cacc_env.py, imported unmodified) in the authoring
container.  The GPU box has no /root/reference, so the vectors are committed.

    python tests/golden/make_golden_env.py

Every case stores: the ini-equivalent parameters, the uniform U drawn at
reset (np.random.seed(seed); np.random.rand()), the action tape and the full
float64 trajectory (h, v, u, reward, global_reward, done, observations).
"""
import configparser
import io
import os
import sys

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE

INI = """
[ENV_CONFIG]
control_interval_sec = 0.1
episode_length_sec = 60
agent = {agent}
batch_size = 60
coop_gamma = {coop_gamma}
headway_min = 1
headway_st = 5
headway_go = 35
speed_max = 30
accel_max = 2.5
accel_min = -2.5
reward_v = 1
reward_u = 0.1
collision_penalty = 1000
headway_target = 20
speed_target = 15
norm_headway = 10
norm_speed = 7.5
n_vehicle = 8
scenario = cacc_{scenario}
seed = {seed}
test_seeds = 10000,20000
"""


def tape(kind, T, N, rng):
    if kind.startswith('const'):
        return np.full((T, N), int(kind[5:]), dtype=np.int64)
    if kind == 'cyclic':  # (step + 3*agent) mod 4
        s = np.arange(T)[:, None]
        a = np.arange(N)[None, :]
        return ((s + 3 * a) % 4).astype(np.int64)
    if kind == 'random':
        return rng.randint(0, 4, size=(T, N)).astype(np.int64)
    if kind == 'mild':   # mostly action 3 with sparse random switches: long non-colliding runs
        x = np.full((T, N), 3, dtype=np.int64)
        m = rng.rand(T, N) < 0.15
        x[m] = rng.randint(0, 4, size=m.sum())
        return x
    raise ValueError(kind)


def run_case(name, scenario, agent, seed, kind, train_mode=True, coop_gamma=-1, test_ind=-1):
    sys.path.insert(0, REF)
    from envs.cacc_env import CACCEnv  # the reference, unmodified
    cp = configparser.ConfigParser()
    cp.read_file(io.StringIO(INI.format(agent=agent, scenario=scenario, seed=seed, coop_gamma=coop_gamma)))
    env = CACCEnv(cp['ENV_CONFIG'])
    env.train_mode = train_mode
    rng = np.random.RandomState(1234 + seed)
    T, N = env.T, env.n_agent
    acts = tape(kind, T, N, rng)
    # which seed will reset() use?  cacc_env.py:169-176
    used_seed = env.seed if train_mode else (env.seed - 1 if test_ind < 0 else env.test_seeds[test_ind])
    np.random.seed(used_seed)
    U = np.random.rand()
    ob = env.reset(test_ind=test_ind)
    fps = rng.dirichlet(np.ones(4), size=(T + 1, N))  # synthetic fingerprints for ia2c_fp
    if agent == 'ia2c_fp':
        env.update_fingerprint(fps[0])
        ob = env._get_state()
    n_s = [len(o) for o in ob]
    obs = np.zeros((T + 1, N, max(n_s)))
    for i, o in enumerate(ob):
        obs[0, i, :len(o)] = o
    hs, vs, us = [env.hs_cur.copy()], [env.vs_cur.copy()], [env.us_cur.copy()]
    rew, grew, dones = [], [], []
    v0s = env.v0s.copy()
    steps = 0
    for t in range(T):
        if agent == 'ia2c_fp':
            env.update_fingerprint(fps[t + 1])
        ob, r, d, g = env.step(acts[t])
        steps += 1
        for i, o in enumerate(ob):
            obs[t + 1, i, :len(o)] = o
        hs.append(np.array(env.hs_cur, dtype=np.float64))
        vs.append(np.array(env.vs_cur, dtype=np.float64))
        us.append(np.array(env.us_cur, dtype=np.float64))
        rew.append(np.broadcast_to(np.asarray(r, dtype=np.float64), (N,)).copy())
        grew.append(g)
        dones.append(d)
        if d:
            break
    out = dict(U=U, used_seed=used_seed, acts=acts[:steps], h=np.array(hs), v=np.array(vs), u=np.array(us),
               reward=np.array(rew), global_reward=np.array(grew), done=np.array(dones),
               obs=obs[:steps + 1], n_s=np.array(n_s), v0s=v0s, fps=fps[:steps + 1],
               scenario=scenario, agent=agent, seed=seed, train_mode=train_mode,
               coop_gamma=coop_gamma, neighbor_mask=env.neighbor_mask, distance_mask=env.distance_mask)
    np.savez_compressed(os.path.join(OUT, 'cacc_%s.npz' % name), **out)
    print('%-28s steps=%3d collided=%s sum_g=%.10f' % (name, steps, env.collision, float(np.sum(grew))))


if __name__ == '__main__':
    if '--out' in sys.argv:                       # regeneration check (tests/test_golden_regen.py): write elsewhere
        OUT = sys.argv[sys.argv.index('--out') + 1]
    run_case('catchup_nc_const0', 'catchup', 'ma2c_nc', 12, 'const0')
    run_case('catchup_nc_const1', 'catchup', 'ma2c_nc', 12, 'const1')       # collides, ends at step 120
    run_case('catchup_nc_const3', 'catchup', 'ma2c_nc', 12, 'const3')
    run_case('catchup_nc_cyclic', 'catchup', 'ma2c_nc', 12, 'cyclic')
    run_case('catchup_nc_mild', 'catchup', 'ma2c_nc', 13, 'mild')
    run_case('catchup_ia2c_mild', 'catchup', 'ia2c', 14, 'mild')
    run_case('catchup_fp_mild', 'catchup', 'ia2c_fp', 15, 'mild')
    run_case('catchup_nc_random', 'catchup', 'ma2c_nc', 16, 'random')
    run_case('slowdown_nc_const3', 'slowdown', 'ma2c_nc', 12, 'const3')
    run_case('slowdown_nc_mild', 'slowdown', 'ma2c_nc', 17, 'mild')
    run_case('slowdown_nc_random', 'slowdown', 'ma2c_nc', 18, 'random')
    run_case('slowdown_ia2c_cyclic', 'slowdown', 'ia2c', 19, 'cyclic')
    run_case('catchup_nc_test_mild', 'catchup', 'ma2c_nc', 20, 'mild', train_mode=False)
    run_case('slowdown_nc_spatial_mild', 'slowdown', 'ma2c_nc', 21, 'mild', coop_gamma=0.9)
    run_case('catchup_nc_eval_seed', 'catchup', 'ma2c_nc', 22, 'const3', train_mode=False, test_ind=1)
