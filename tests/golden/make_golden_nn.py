"""Generate tests/golden/nn_*.npz by running the REAL reference model code
(/root/reference/agents/{utils,policies,models}.py, unmodified) on the fake-TF
shim oracle/tf1_shim (torch-CPU, float64).  See the shim's docstring for what
this pins and what it does not (TF kernel semantics of RMSProp/clip/softmax).

    python tests/golden/make_golden_nn.py

"Scripted" cases drive `model.forward('p')`, `model.forward('v')`,
`model.add_transition`, `model.backward` in exactly the order of the
reference's Trainer.explore/run (utils.py:163-254) for three n_step batches on
synthetic observations, and record per-step pi / v, the bootstrap R, the loss
and global grad-norm of every update, and per-variable statistics + samples of
the weights after every update.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'tf1_shim'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import tensorflow as tf  # noqa: E402  (the shim)
from agents.models import IA2C, IA2C_FP, IA2C_CU, MA2C_NC, MA2C_IC3, MA2C_DIAL  # noqa: E402  (the reference)
from helpers import cacc_config  # noqa: E402

# tf.gradients is recorded so that run_batched can fetch the RAW (pre-clip) gradient of every optimiser
GRAD_NODES = {}
_orig_gradients = tf.gradients


def _recording_gradients(loss, wts):
    gs = _orig_gradients(loss, wts)
    GRAD_NODES[id(loss)] = (gs, list(wts))
    return gs


tf.gradients = _recording_gradients

CLS = {'ia2c': IA2C, 'ia2c_fp': IA2C_FP, 'ma2c_nc': MA2C_NC, 'ma2c_ic3': MA2C_IC3, 'ma2c_cu': IA2C_CU,
       'ma2c_dial': MA2C_DIAL}
N_SAMPLE = 16


def var_stats(variables):
    """[n_var, 3 + N_SAMPLE]: sum, abs-sum, l2, then samples at idx (k*7919) % size."""
    rows = []
    for v in variables:
        a = (v if isinstance(v, np.ndarray) else v.numpy()).astype(np.float64).ravel()
        idx = (np.arange(N_SAMPLE) * 7919) % a.size
        rows.append(np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], a[idx]]))
    return np.array(rows)


def line_masks(n):
    idx = np.arange(n)
    d = np.abs(idx[:, None] - idx[None, :])
    return (d == 1).astype(int), d.astype(int)


def grid_masks(side=5):
    """5x5 grid neighbourhood / hop distance (large_grid_env.py:58-105 semantics: 4-neighbour lattice)."""
    n = side * side
    nb = np.zeros((n, n), dtype=int)
    dist = np.zeros((n, n), dtype=int)
    for i in range(n):
        for j in range(n):
            dist[i, j] = abs(i // side - j // side) + abs(i % side - j % side)
    nb[dist == 1] = 1
    return nb, dist


def ragged_graph():
    """5 heterogeneous agents (identical=False nets): own observation widths / action counts differ,
    the neighbourhood is irregular and agent 4 is isolated (no fingerprint / message inputs at all, like Monaco's
    '8996', real_net_env.py:24)."""
    n = 5
    nb = np.zeros((n, n), dtype=int)
    for i, j in [(0, 1), (1, 2), (2, 3), (0, 2)]:
        nb[i, j] = nb[j, i] = 1
    dist = np.full((n, n), n, dtype=int)                 # hop distance; unreachable pairs: n (never used: coop_gamma < 0)
    for i in range(n):
        dist[i, i] = 0
    for i in range(n):
        for j in range(n):
            if nb[i, j]:
                dist[i, j] = 1
    for k in range(n):
        for i in range(n):
            for j in range(n):
                dist[i, j] = min(dist[i, j], dist[i, k] + dist[k, j])
    return nb, dist, [3, 5, 4, 6, 2], [2, 3, 4, 2, 5]


def run_ragged(name, agent, seed, n_step, n_batch=3):
    """The scripted three-batch run of run_scripted for the reference's heterogeneous nets (identical=False:
    agents/utils.py:220-341, 420-512, 602-719; policies.py:59-77 with na_dim_ls; models.py:118-131, 229-244)."""
    cp = cacc_config(agent=agent, n_step=n_step, reward_norm=50.0, coop_gamma=-1)
    mc = cp['MODEL_CONFIG']
    nb, dist, n_own, n_a_ls = ragged_graph()
    N, F, A = len(n_own), max(n_own), max(n_a_ls)
    is_ma = agent.startswith('ma2c')
    nbr = [np.where(nb[i] == 1)[0] for i in range(N)]
    # IA2C_FP adds the neighbours' fingerprint widths itself (models.py:176-187)
    n_s_ls = list(n_own) if is_ma else [n_own[i] + sum(n_own[j] for j in nbr[i]) for i in range(N)]
    np.random.seed(seed)
    model = CLS[agent](n_s_ls, n_a_ls, nb, dist, -1, 10000, mc, seed=seed)
    assert not model.identical_agent
    variables = tf.global_variables()
    policies = model.policy if isinstance(model.policy, list) else [model.policy]
    train_to_pol = {id(p._train): p for p in policies}
    train_log = []
    orig_run = tf.Session.run

    def logging_run(self, fetches, feed_dict=None):
        if isinstance(fetches, list):
            for f in fetches:
                if getattr(f, 'is_train', False):
                    p = train_to_pol[id(f)]
                    train_log.append(orig_run(self, [p.loss, p.grad_norm], feed_dict))
        return orig_run(self, fetches, feed_dict)
    tf.Session.run = logging_run

    rng = np.random.RandomState(seed + 1000)
    X = rng.normal(0, 0.7, size=(n_batch, n_step + 1, N, F))
    for i in range(N):
        X[:, :, i, n_own[i]:] = 0.0
    ACT = np.stack([rng.randint(0, n_a_ls[i], size=(n_batch, n_step + 1)) for i in range(N)], axis=-1)
    REW = rng.normal(-30, 20, size=(n_batch, n_step, 1))
    out = dict(stats0=var_stats(variables), names=np.array([v.full_name for v in variables]),
               shapes=np.array([str(tuple(v.value.shape)) for v in variables]))
    PI = np.zeros((n_batch, n_step + 1, N, A))
    V = np.zeros((n_batch, n_step + 1, N))
    RB = np.zeros((n_batch, N))
    STATS, LOSS, STATES = [], [], []
    uniform = lambda: [np.ones(n_a_ls[i]) / n_a_ls[i] for i in range(N)]          # noqa: E731
    fp = uniform()

    def make_ob(x):
        ob = []
        for i in range(N):
            cur = [x[i, :n_own[i]]]
            if not is_ma:
                cur += [x[j, :n_own[j]] for j in nbr[i]]
            if agent == 'ia2c_fp':
                cur += [fp[j] for j in nbr[i]]
            ob.append(np.concatenate(cur))
        return ob

    def get_policy(ob, done):
        pi = model.forward(ob, done, fp) if is_ma else model.forward(ob, done)
        return [np.asarray(p, dtype=np.float64).reshape(-1) for p in pi]

    def get_value(ob, done, ps, action):
        if is_ma:
            return np.array(model.forward(ob, done, ps, np.array(action), 'v')).reshape(N), ps
        na = [action[nb[i] == 1] for i in range(N)]
        return np.array(model.forward(ob, done, na, 'v'), dtype=np.float64).reshape(N), na

    def pad(pi):
        o = np.zeros((N, A))
        for i in range(N):
            o[i, :n_a_ls[i]] = pi[i]
        return o

    done = True
    model.reset()
    for b in range(n_batch):
        if done:
            model.reset()
            fp = uniform()
        for t in range(n_step):
            ob = make_ob(X[b, t])
            ps = [f.copy() for f in fp]
            pi = get_policy(ob, done)
            a = ACT[b, t]
            v, extra = get_value(ob, done, ps, a)
            fp = [p.copy() for p in pi]
            done = (b == 1 and t == n_step - 1)
            model.add_transition(ob, extra, a, float(REW[b, t, 0]), v, done)
            PI[b, t], V[b, t] = pad(pi), v
        if done:
            R = np.zeros(N)
        else:
            ob = make_ob(X[b, n_step])
            ps = [f.copy() for f in fp]
            pi = get_policy(ob, done)
            R, _ = get_value(ob, done, ps, ACT[b, n_step])
            PI[b, n_step], V[b, n_step] = pad(pi), R
        RB[b] = R
        k0 = len(train_log)
        model.backward(R, 0)
        LOSS.append(np.array(train_log[k0:], dtype=np.float64))
        STATS.append(var_stats(variables))
        sf = [p.states_fw for p in policies]
        STATES.append(np.array(sf, dtype=np.float64).reshape(N, -1))
    tf.Session.run = orig_run
    out.update(X=X, ACT=ACT, REW=REW, PI=PI, V=V, RB=RB, LOSS=np.array(LOSS), STATS=np.array(STATS),
               STATES=np.array(STATES), nb=nb, dist=dist, agent=agent, topo='ragged', seed=seed, n_step=n_step,
               coop_gamma=-1, reward_norm=50.0, n_own=np.array(n_own), n_a_ls=np.array(n_a_ls), n_s_ls=np.array(n_s_ls))
    np.savez_compressed(os.path.join(OUT, 'nn_%s.npz' % name), **out)
    nparam = sum(int(np.prod(v.value.shape)) for v in variables)
    print('%-22s params=%7d loss=%s gnorm=%s' % (name, nparam, np.round(LOSS[0][:, 0], 5)[:3],
                                                 np.round(LOSS[0][:, 1], 4)[:3]))


RAGGED = [('ia2c_ragged', 'ia2c', 30), ('ia2c_fp_ragged', 'ia2c_fp', 31), ('ma2c_nc_ragged', 'ma2c_nc', 32),
          ('ma2c_ic3_ragged', 'ma2c_ic3', 33), ('ma2c_cu_ragged', 'ma2c_cu', 34), ('ma2c_dial_ragged', 'ma2c_dial', 35)]


def run_scripted(name, agent, topo, seed, n_step, n_batch=3, coop_gamma=-1):
    cp = cacc_config(agent=agent, n_step=n_step, reward_norm=50.0, coop_gamma=coop_gamma)
    mc = cp['MODEL_CONFIG']
    if topo == 'line':
        N, n_feat, A = 8, 5, 4
        nb, dist = line_masks(N)
    else:
        N, n_feat, A = 25, 12, 5
        nb, dist = grid_masks(5)
    is_ma = agent.startswith('ma2c')
    nbr = [np.where(nb[i] == 1)[0] for i in range(N)]
    n_s_ls = [n_feat if is_ma else n_feat * (1 + len(nbr[i])) for i in range(N)]
    n_a_ls = [A] * N
    np.random.seed(seed)                      # cacc_env.py:22 -- weight init consumes this stream
    model = CLS[agent](n_s_ls, n_a_ls, nb, dist, coop_gamma, 10000, mc, seed=seed)
    variables = tf.global_variables()
    policies = model.policy if isinstance(model.policy, list) else [model.policy]
    train_to_pol = {id(p._train): p for p in policies}
    train_log = []
    orig_run = tf.Session.run

    def logging_run(self, fetches, feed_dict=None):
        if isinstance(fetches, list):
            for f in fetches:
                if getattr(f, 'is_train', False):
                    p = train_to_pol[id(f)]
                    train_log.append(orig_run(self, [p.loss, p.grad_norm], feed_dict))
        return orig_run(self, fetches, feed_dict)
    tf.Session.run = logging_run

    rng = np.random.RandomState(seed + 1000)
    X = rng.normal(0, 0.7, size=(n_batch, n_step + 1, N, n_feat))
    ACT = rng.randint(0, A, size=(n_batch, n_step + 1, N))
    REW = rng.normal(-30, 20, size=(n_batch, n_step, N if coop_gamma >= 0 else 1))
    out = dict(stats0=var_stats(variables), names=np.array([v.full_name for v in variables]),
               shapes=np.array([str(tuple(v.value.shape)) for v in variables]))
    PI = np.zeros((n_batch, n_step + 1, N, A))
    V = np.zeros((n_batch, n_step + 1, N))
    RB = np.zeros((n_batch, N))
    DONE0 = np.zeros(n_batch, dtype=bool)
    STATS, LOSS, STATES = [], [], []
    fp = np.ones((N, A)) / A

    def make_ob(x):
        ob = []
        for i in range(N):
            cur = [x[i]]
            if not is_ma:
                cur += [x[j] for j in nbr[i]]
            if agent == 'ia2c_fp':
                cur += [fp[j] for j in nbr[i]]
            ob.append(np.concatenate(cur))
        return ob

    def get_policy(ob, done):
        if is_ma:
            return np.array(model.forward(ob, done, fp))
        return np.array(model.forward(ob, done))

    def get_value(ob, done, ps, action):
        if is_ma:
            return np.array(model.forward(ob, done, ps, np.array(action), 'v'))
        na = [action[nb[i] == 1] for i in range(N)]
        return np.array(model.forward(ob, done, na, 'v')), na

    # episode layout: batch 0 starts an episode, batch 1 continues it and ends it, batch 2 starts anew
    done = True
    model.reset()
    for b in range(n_batch):
        DONE0[b] = done
        if done:
            model.reset()
            fp = np.ones((N, A)) / A
        for t in range(n_step):
            ob = make_ob(X[b, t])
            ps = fp.copy()
            pi = get_policy(ob, done)
            a = ACT[b, t]
            if is_ma:
                v = get_value(ob, done, ps, a)
                extra = ps
            else:
                v, extra = get_value(ob, done, None, a)
            fp = pi.copy()                                   # env.update_fingerprint(policy)
            r = REW[b, t] if coop_gamma >= 0 else float(REW[b, t, 0])
            done = (b == 1 and t == n_step - 1)              # the episode ends with batch 1
            model.add_transition(ob, extra, a, r, v, done)
            PI[b, t], V[b, t] = pi, v
        if done:
            R = np.zeros(N)
        else:                                                # bootstrap, utils.py:192-196
            ob = make_ob(X[b, n_step])
            ps = fp.copy()
            pi = get_policy(ob, done)
            a = ACT[b, n_step]
            res = get_value(ob, done, ps, a)
            R = res if is_ma else res[0]
            PI[b, n_step], V[b, n_step] = pi, R
        RB[b] = R
        k0 = len(train_log)
        model.backward(R, 0)
        LOSS.append(np.array(train_log[k0:], dtype=np.float64))
        STATS.append(var_stats(variables))
        sf = [p.states_fw for p in policies]
        STATES.append(np.array(sf, dtype=np.float64).reshape(N, -1))
    tf.Session.run = orig_run
    out.update(X=X, ACT=ACT, REW=REW, PI=PI, V=V, RB=RB, DONE0=DONE0, LOSS=np.array(LOSS),
               STATS=np.array(STATS), STATES=np.array(STATES), nb=nb, dist=dist, agent=agent,
               topo=topo, seed=seed, n_step=n_step, coop_gamma=coop_gamma, reward_norm=50.0)
    np.savez_compressed(os.path.join(OUT, 'nn_%s.npz' % name), **out)
    nparam = sum(int(np.prod(v.value.shape)) for v in variables)
    print('%-22s params=%7d loss=%s gnorm=%s' % (name, nparam, np.round(LOSS[0][:, 0], 5)[:3],
                                                 np.round(LOSS[0][:, 1], 4)[:3]))


def run_batched(name, agent, topo, seed, n_step, K=4):
    """The E > 1 update contract: K independent replicas of the reference model with IDENTICAL
    weights each roll out one n_step batch (models.py:26-51 / 198-227, Trainer.explore order) and run `backward`
    (models.py:34-42, 211-215) -- with the optimiser step intercepted, so the weights stay put and the raw gradient
    of every optimiser (policies.py:32-33, 257-258) is recorded.  The batched update the product performs on E = K
    lock-stepped replicas must equal: mean over replicas of those gradients -> clip_by_global_norm -> ONE RMSProp
    step (policies.py:34-39, 259-264; loss means over T, so the mean over T*E is the mean of the replica means).

    Episode layout (what BatchedTrainer produces around a batch boundary): every replica first runs a PREFIX batch
    (no update); replicas 0, 1 end their episode there and start the main batch fresh (done = True, zero state,
    uniform fingerprints -- the reference's env.reset(); model.reset()), replicas 2.. continue (done = False, states
    and fingerprints carried over, states_bw <- states_fw as in policies.py:115, 211).  Replica 1's episode ends with
    the last step of the main batch (R = 0, utils.py:189-190), the others bootstrap (utils.py:192-196)."""
    cp = cacc_config(agent=agent, n_step=n_step, reward_norm=50.0, coop_gamma=-1)
    mc = cp['MODEL_CONFIG']
    if topo == 'line':
        N, n_feat, A = 8, 5, 4
        nb, dist = line_masks(N)
    else:
        N, n_feat, A = 25, 12, 5
        nb, dist = grid_masks(5)
    is_ma = agent.startswith('ma2c')
    nbr = [np.where(nb[i] == 1)[0] for i in range(N)]
    n_s_ls = [n_feat if is_ma else n_feat * (1 + len(nbr[i])) for i in range(N)]
    np.random.seed(seed)
    model = CLS[agent](n_s_ls, [A] * N, nb, dist, -1, 10 ** 9, mc, seed=seed)
    variables = tf.global_variables()
    by_name = {v.full_name: v for v in variables}
    policies = model.policy if isinstance(model.policy, list) else [model.policy]
    train_to_pol = {id(p._train): p for p in policies}
    grad_log = []
    orig_run = tf.Session.run

    def intercept(self, fetches, feed_dict=None):
        if isinstance(fetches, list) and any(getattr(f, 'is_train', False) for f in fetches):
            for f in fetches:
                if getattr(f, 'is_train', False):
                    p = train_to_pol[id(f)]
                    gs, wts = GRAD_NODES[id(p.loss)]
                    vals = orig_run(self, [p.loss] + list(gs), feed_dict)
                    grad_log.append((p, float(vals[0]), [np.array(g, dtype=np.float64) for g in vals[1:]],
                                     [w.full_name for w in wts]))
            return [None] * len(fetches)                      # the optimiser step is NOT applied
        if any(fetches is getattr(p, '_consensus_update', None) for p in policies):
            return None           # ConseNet's neighbourhood averaging (policies.py:351-355) follows the (skipped) step: applied once below
        return orig_run(self, fetches, feed_dict)
    tf.Session.run = intercept

    T = n_step
    rng = np.random.RandomState(seed + 2000)
    X = rng.normal(0, 0.7, size=(K, 2, T + 1, N, n_feat))
    ACT = rng.randint(0, A, size=(K, 2, T + 1, N))
    REW = rng.normal(-30, 20, size=(K, 2, T))
    PI = np.zeros((K, 2, T + 1, N, A))
    V = np.zeros((K, 2, T + 1, N))
    RB = np.zeros((K, N))
    STATES = np.zeros((K, N, 2 * int(mc['num_lstm'])))
    state = {'fp': None}

    def make_ob(x):
        ob = []
        for i in range(N):
            cur = [x[i]]
            if not is_ma:
                cur += [x[j] for j in nbr[i]]
            if agent == 'ia2c_fp':
                cur += [state['fp'][j] for j in nbr[i]]
            ob.append(np.concatenate(cur))
        return ob

    def decide(x, done, a, update_fp=True):
        ob = make_ob(x)
        ps = state['fp'].copy()
        if is_ma:
            pi = np.array(model.forward(ob, done, ps))
            v = np.array(model.forward(ob, done, ps, np.array(a), 'v'))
            extra = ps
        else:
            pi = np.array(model.forward(ob, done))
            extra = [a[nb[i] == 1] for i in range(N)]
            v = np.array(model.forward(ob, done, extra, 'v'))
        if update_fp:
            state['fp'] = pi.copy()                           # env.update_fingerprint (utils.py:173), not at the bootstrap
        return ob, extra, pi, v

    def batch(k, ph, done, last_done):
        for t in range(T):
            a = ACT[k, ph, t]
            ob, extra, pi, v = decide(X[k, ph, t], done, a)
            done = bool(last_done and t == T - 1)
            model.add_transition(ob, extra, a, float(REW[k, ph, t]), v, done)
            PI[k, ph, t], V[k, ph, t] = pi, v
        if done:
            R = np.zeros(N)
        else:
            _, _, pi, R = decide(X[k, ph, T], done, ACT[k, ph, T], update_fp=False)
            PI[k, ph, T], V[k, ph, T] = pi, R
        model.backward(R, 0)                                  # intercepted: gradients recorded, states_bw <- states_fw
        return R

    per_replica = []
    for k in range(K):
        model.reset()
        state['fp'] = np.ones((N, A)) / A
        if k >= 2:                                            # continuing replicas: a prefix batch, discarded
            batch(k, 0, True, False)
            grad_log.clear()
            RB[k] = batch(k, 1, False, False)
        else:
            RB[k] = batch(k, 1, True, k == 1)
        per_replica.append(list(grad_log))
        grad_log.clear()
        STATES[k] = np.array([p.states_fw for p in policies], dtype=np.float64).reshape(N, -1)
    tf.Session.run = orig_run

    # ---- expected batched update: mean gradient -> clip -> RMSProp from fresh slots (ms = 1), per optimiser
    lr, decay, eps, clip = float(mc['lr_init']), float(mc['rmsp_alpha']), float(mc['rmsp_epsilon']), float(mc['max_grad_norm'])
    new_w = {v.full_name: v.numpy().astype(np.float64) for v in variables}
    mean_g = {}
    n_opt = len(per_replica[0])
    LOSSK = np.array([[per_replica[k][o][1] for o in range(n_opt)] for k in range(K)])
    GN = np.zeros(n_opt)
    for o in range(n_opt):
        names = per_replica[0][o][3]
        gs = [np.mean([per_replica[k][o][2][j] for k in range(K)], axis=0) for j in range(len(names))]
        GN[o] = np.sqrt(sum((g * g).sum() for g in gs))
        scale = clip * min(1.0 / GN[o], 1.0 / clip)
        for nm, g in zip(names, gs):
            g = g * scale
            ms = 1.0 + (g * g - 1.0) * (1.0 - decay)
            new_w[nm] = by_name[nm].numpy().astype(np.float64) - lr * g / np.sqrt(ms + eps)
            mean_g[nm] = g / scale
    if agent == 'ma2c_cu':
        # the consensus update that follows the optimiser step (policies.py:351-364, 403-426): every agent's LSTM variables
        # <- mean over itself and its neighbours, all sources read before any target is written (the shim's tf.group)
        stepped = dict(new_w)
        for i in range(N):
            for var in ('wx', 'wh', 'b'):
                new_w['cu/lstm_%da/%s' % (i, var)] = np.mean([stepped['cu/lstm_%da/%s' % (j, var)] for j in [i] + list(nbr[i])], axis=0)
    out = dict(stats0=var_stats(variables), names=np.array([v.full_name for v in variables]),
               shapes=np.array([str(tuple(v.value.shape)) for v in variables]),
               X=X, ACT=ACT, REW=REW, PI=PI, V=V, RB=RB, STATES=STATES, LOSSK=LOSSK, LOSS=LOSSK.mean(0), GN=GN,
               STATS=var_stats([new_w[v.full_name] for v in variables]),
               GSTATS=var_stats([mean_g.get(v.full_name, np.zeros(tuple(v.value.shape))) for v in variables]),
               nb=nb, dist=dist, agent=agent, topo=topo, seed=seed, n_step=n_step, K=K, coop_gamma=-1, reward_norm=50.0)
    np.savez_compressed(os.path.join(OUT, 'nnb_%s.npz' % name), **out)
    print('%-22s K=%d T=%d loss=%s gnorm=%s' % ('nnb_' + name, K, T, np.round(out['LOSS'], 5)[:3], np.round(GN, 4)[:3]))


BATCHED = [('ia2c_fp_line', 'ia2c_fp', 'line', 40, 60), ('ma2c_nc_line', 'ma2c_nc', 'line', 41, 60),
           ('ma2c_ic3_grid', 'ma2c_ic3', 'grid', 42, 120),
           # round 4: the algorithms whose batched update was pinned at E = 1 only, and NeurComm on the grid (message input
           # 64 x 4 = 256 wide: the product's separate-launch message path, not the step kernel's pre-phase)
           ('ia2c_line', 'ia2c', 'line', 43, 60), ('ma2c_cu_line', 'ma2c_cu', 'line', 44, 60),
           ('ma2c_dial_line', 'ma2c_dial', 'line', 45, 60), ('ma2c_nc_grid', 'ma2c_nc', 'grid', 46, 30)]


def run_ortho():
    from agents.utils import ortho_init
    np.random.seed(12)
    shapes = [(15, 64), (64, 256), (128, 256), (64, 4), (72, 1), (192, 256), (8, 64), (1, 3)]
    out = {}
    for k, s in enumerate(shapes):
        out['w%d' % k] = ortho_init()(s, None)
    out['shapes'] = np.array(shapes)
    out['after'] = np.random.rand()
    np.savez_compressed(os.path.join(OUT, 'ortho_init.npz'), **out)
    print('ortho w0[0,0] = %.7f' % out['w0'][0, 0])


def main(argv):
    """No arguments: regenerate EVERY fixture.  --only a,b,...: the named fixtures (file names without .npz)."""
    only = None
    if '--only' in argv:
        only = set(argv[argv.index('--only') + 1].split(','))

    def want(fname):
        return only is None or fname in only
    if want('ortho_init'):
        run_ortho()
    for nm, ag, topo, sd, ns, cg in SCRIPTED:
        if want('nn_' + nm):
            tf.reset_default_graph()
            run_scripted(nm, ag, topo, sd, ns, coop_gamma=cg)
    for nm, ag, sd in RAGGED:
        if want('nn_' + nm):
            tf.reset_default_graph()
            run_ragged(nm, ag, sd, 4)
    for nm, ag, topo, sd, ns in BATCHED:
        if want('nnb_' + nm):
            tf.reset_default_graph()
            run_batched(nm, ag, topo, sd, ns)


SCRIPTED = [('ia2c_line', 'ia2c', 'line', 12, 6, -1), ('ia2c_fp_line', 'ia2c_fp', 'line', 13, 6, -1),
            ('ma2c_nc_line', 'ma2c_nc', 'line', 14, 6, -1), ('ma2c_ic3_line', 'ma2c_ic3', 'line', 15, 6, -1),
            ('ma2c_ic3_grid', 'ma2c_ic3', 'grid', 16, 4, -1), ('ma2c_nc_grid', 'ma2c_nc', 'grid', 17, 4, -1),
            ('ma2c_nc_line_spatial', 'ma2c_nc', 'line', 18, 6, 0.9), ('ia2c_line_spatial', 'ia2c', 'line', 19, 6, 0.8),
            ('ma2c_cu_line', 'ma2c_cu', 'line', 20, 6, -1), ('ma2c_dial_line', 'ma2c_dial', 'line', 21, 6, -1),
            ('ma2c_cu_grid', 'ma2c_cu', 'grid', 22, 4, -1), ('ma2c_dial_grid', 'ma2c_dial', 'grid', 23, 4, -1)]


if __name__ == '__main__':
    if '--out' in sys.argv:                       # regeneration check (tests/test_golden_regen.py): write elsewhere
        OUT = sys.argv[sys.argv.index('--out') + 1]
    main(sys.argv[1:])
