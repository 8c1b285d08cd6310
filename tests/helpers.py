"""Shared test helpers (CPU-safe)."""
import configparser
import io
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CACC_INI = """
[MODEL_CONFIG]
rmsp_alpha = 0.99
rmsp_epsilon = 1e-5
max_grad_norm = 40
gamma = 0.99
lr_init = 5e-4
lr_decay = constant
entropy_coef = 0.05
value_coef = 0.5
num_lstm = 64
num_fc = 64
batch_size = {n_step}
reward_norm = {reward_norm}
reward_clip = -1

[TRAIN_CONFIG]
total_step = {total_step}
test_interval = 2e6
log_interval = 1e4

[ENV_CONFIG]
control_interval_sec = 0.1
episode_length_sec = 60
agent = {agent}
batch_size = {n_step}
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


def cacc_config(agent='ma2c_nc', scenario='catchup', seed=12, coop_gamma=-1, n_step=60,
                reward_norm=5000.0, total_step=1200):
    cp = configparser.ConfigParser()
    cp.read_file(io.StringIO(CACC_INI.format(agent=agent, scenario=scenario, seed=seed,
                                             coop_gamma=coop_gamma, n_step=n_step,
                                             reward_norm=reward_norm, total_step=total_step)))
    return cp


def load_npz(path):
    with np.load(path, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


# --------------------------------------------------------------------------- NN golden driver
N_SAMPLE = 16


def var_stats_from_named(named):
    rows = []
    for _, a in named:
        a = np.asarray(a, dtype=np.float64).ravel()
        idx = (np.arange(N_SAMPLE) * 7919) % a.size
        rows.append(np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], a[idx]]))
    return np.array(rows)


def build_product_model(z, device):
    """Instantiate the product model exactly like tests/golden/make_golden_nn.py built the reference one."""
    from deeprl_network_amd.agents import models
    agent, topo = str(z['agent']), str(z['topo'])
    cls = {'ia2c': models.IA2C, 'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3,
           'ma2c_cu': models.IA2C_CU, 'ma2c_dial': models.MA2C_DIAL}[agent]
    n_step, coop_gamma, seed = int(z['n_step']), float(z['coop_gamma']), int(z['seed'])
    cp = cacc_config(agent=agent, n_step=n_step, reward_norm=float(z['reward_norm']), coop_gamma=coop_gamma)
    nb, dist = z['nb'], z['dist']
    N = nb.shape[0]
    np.random.seed(seed)
    if topo == 'ragged':        # heterogeneous agents: own widths / action counts recorded by make_golden_nn.run_ragged
        return cls([int(x) for x in z['n_s_ls']], [int(x) for x in z['n_a_ls']], nb, dist, coop_gamma, 10000,
                   cp['MODEL_CONFIG'], seed=seed, num_envs=1, device=device, n_feat_ls=[int(x) for x in z['n_own']])
    n_feat, A = (5, 4) if topo == 'line' else (12, 5)
    is_ma = agent.startswith('ma2c')
    n_s_ls = [n_feat if is_ma else n_feat * (1 + int(nb[i].sum())) for i in range(N)]
    model = cls(n_s_ls, [A] * N, nb, dist, coop_gamma, 10000, cp['MODEL_CONFIG'], seed=seed, num_envs=1,
                device=device)
    return model


def drive_scripted(model, z):
    """Replays the scripted three-batch run of make_golden_nn.run_scripted / run_ragged through the product's
    reference-compatible API and returns the same record (ragged policies zero padded to the widest action set)."""
    agent = str(z['agent'])
    is_ma = agent.startswith('ma2c')
    nb = z['nb']
    N = nb.shape[0]
    A = model.n_a
    nbr = [np.where(nb[i] == 1)[0] for i in range(N)]
    X, ACT, REW = z['X'], z['ACT'], z['REW']
    n_batch, n_step = X.shape[0], int(z['n_step'])
    coop_gamma = float(z['coop_gamma'])
    ragged = str(z['topo']) == 'ragged'
    n_own = [int(v) for v in z['n_own']] if ragged else [X.shape[-1]] * N
    n_a_ls = [int(v) for v in z['n_a_ls']] if ragged else [A] * N
    PI = np.zeros_like(z['PI'])
    V = np.zeros_like(z['V'])
    RB = np.zeros_like(z['RB'])
    STATS, LOSS, GN, STATES = [], [], [], []
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

    def pad(pi):
        o = np.zeros((N, A))
        for i in range(N):
            o[i, :n_a_ls[i]] = np.asarray(pi[i]).reshape(-1)
        return o

    def as_list(ps):
        return ps if ragged else np.array(ps)

    done = True
    model.reset()
    for b in range(n_batch):
        if done:
            model.reset()
            fp = uniform()
        for t in range(n_step):
            ob = make_ob(X[b, t])
            ps = [f.copy() for f in fp]
            a = ACT[b, t]
            if is_ma:
                pi = model.forward(ob, done, as_list(fp))
                v = np.array(model.forward(ob, done, as_list(ps), np.array(a), 'v'))
                extra = as_list(ps)
            else:
                pi = model.forward(ob, done)
                extra = [a[nb[i] == 1] for i in range(N)]
                v = np.array(model.forward(ob, done, extra, 'v'))
            fp = [np.asarray(p, dtype=np.float64).reshape(-1).copy() for p in pi]
            r = REW[b, t] if coop_gamma >= 0 else float(REW[b, t, 0])
            done = (b == 1 and t == n_step - 1)
            model.add_transition(ob, extra, a, r, v, done)
            PI[b, t], V[b, t] = pad(pi), v
        if done:
            R = np.zeros(N)
        else:
            ob = make_ob(X[b, n_step])
            ps = [f.copy() for f in fp]
            a = ACT[b, n_step]
            if is_ma:
                pi = model.forward(ob, done, as_list(fp))
                R = np.array(model.forward(ob, done, as_list(ps), np.array(a), 'v'))
            else:
                pi = model.forward(ob, done)
                R = np.array(model.forward(ob, done, [a[nb[i] == 1] for i in range(N)], 'v'))
            PI[b, n_step], V[b, n_step] = pad(pi), R
        RB[b] = R
        model.backward(R, 0)
        tot = model.last_loss[3].cpu().numpy().astype(np.float64)
        gn = model.grad_norm.cpu().numpy().astype(np.float64)
        if model.per_agent_optimizer:
            LOSS.append(np.stack([tot, gn], axis=1))
        else:
            LOSS.append(np.array([[tot.sum(), gn[0]]]))
        STATS.append(var_stats_from_named(model.policy.params.ref_variables()))
        STATES.append(np.concatenate([model.c_fw[:, 0].cpu().numpy(), model.h_fw[:, 0].cpu().numpy()], axis=1))
    return dict(PI=PI, V=V, RB=RB, LOSS=np.array(LOSS), STATS=np.array(STATS), STATES=np.array(STATES))


def compare_scripted(out, z, rtol_fw=1e-4, rtol_w=1e-3):
    """NN forward rtol 1e-4; post-update weights rtol 1e-3."""
    np.testing.assert_allclose(out['PI'], z['PI'], rtol=rtol_fw, atol=1e-6, err_msg='pi')
    np.testing.assert_allclose(out['V'], z['V'], rtol=rtol_fw, atol=2e-5, err_msg='v')
    np.testing.assert_allclose(out['RB'], z['RB'], rtol=rtol_fw, atol=2e-5, err_msg='R bootstrap')
    np.testing.assert_allclose(out['STATES'], z['STATES'], rtol=rtol_fw, atol=2e-6, err_msg='states_fw')
    np.testing.assert_allclose(out['LOSS'][..., 0], z['LOSS'][..., 0], rtol=1e-4, atol=1e-5, err_msg='loss')
    np.testing.assert_allclose(out['LOSS'][..., 1], z['LOSS'][..., 1], rtol=1e-4, atol=1e-6, err_msg='grad norm')
    # weights: sums / l2 norms and 16 samples per variable after each update
    s, g = out['STATS'], z['STATS']
    np.testing.assert_allclose(s[..., 1:3], g[..., 1:3], rtol=rtol_w, err_msg='|w| / l2')
    np.testing.assert_allclose(s[..., 3:], g[..., 3:], rtol=rtol_w, atol=2e-6, err_msg='weight samples')


def grid_config(agent='ma2c_ic3', coop_gamma=-1, seed=12, n_step=120):
    cp = configparser.ConfigParser()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'config', 'config_ma2c_cnet_grid.ini')
    cp.read(path)
    cp['ENV_CONFIG']['agent'] = agent
    cp['ENV_CONFIG']['coop_gamma'] = str(coop_gamma)
    cp['ENV_CONFIG']['seed'] = str(seed)
    cp['MODEL_CONFIG']['batch_size'] = str(n_step)
    return cp


def net_config(agent='ma2c_nc', coop_gamma=0.9, seed=12, n_step=120):
    cp = configparser.ConfigParser()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'config', 'config_ma2c_nc_net.ini')
    cp.read(path)
    cp['ENV_CONFIG']['agent'] = agent
    cp['ENV_CONFIG']['coop_gamma'] = str(coop_gamma)
    cp['ENV_CONFIG']['seed'] = str(seed)
    cp['MODEL_CONFIG']['batch_size'] = str(n_step)
    return cp


# --------------------------------------------------------------------------- batched (E = K) update golden
def build_product_batched(z, device):
    """The product model with E = K replicas, initial weights drawn like make_golden_nn.run_batched's reference model."""
    from deeprl_network_amd.agents import models
    agent, topo = str(z['agent']), str(z['topo'])
    cls = {'ia2c': models.IA2C, 'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3,
           'ma2c_cu': models.IA2C_CU, 'ma2c_dial': models.MA2C_DIAL}[agent]
    n_step, seed, K = int(z['n_step']), int(z['seed']), int(z['K'])
    cp = cacc_config(agent=agent, n_step=n_step, reward_norm=float(z['reward_norm']), coop_gamma=-1)
    nb, dist = z['nb'], z['dist']
    N = nb.shape[0]
    n_feat, A = (5, 4) if topo == 'line' else (12, 5)
    is_ma = agent.startswith('ma2c')
    n_s_ls = [n_feat if is_ma else n_feat * (1 + int(nb[i].sum())) for i in range(N)]
    np.random.seed(seed)
    return cls(n_s_ls, [A] * N, nb, dist, -1, 10 ** 9, cp['MODEL_CONFIG'], seed=seed, num_envs=K, device=device)


def drive_batched(model, z, saved=False, compact=False):
    """saved: the rollout hands its activations to the update (model.enable_saved_activations, what BatchedTrainer does
    for uncoupled nets) instead of the update recomputing the forward pass.  compact: the observations are handed over as
    the env's COMPACT slab [K,N,F] (model.enable_compact_obs: the encoders gather the neighbours themselves -- with `saved`
    this is BatchedTrainer's exact configuration, incl. the input encoders inside the lock-step kernel where they fit).

    Replays make_golden_nn.run_batched on the product's BATCHED engine (act / bootstrap / update on E = K
    lock-stepped replicas, the calls BatchedTrainer makes): a prefix batch without update, the episode boundary
    (replicas 0, 1 restart: reset_states(mask), done_pre = 1), the main batch, ONE update.  Observations are written
    into buf_x; the scripted actions are forced through the kernels' own draw with uniforms placed in the middle of the
    golden CDF interval of the wanted action (mode SAMPLE_UNIFORM), so the fused policy/value kernels run unchanged."""
    import torch
    from deeprl_network_amd import ops
    X, ACT, REW, PIg = z['X'], z['ACT'], z['REW'], z['PI']
    K, T = int(z['K']), int(z['n_step'])
    N, A, dev = model.n_agent, model.n_a, model.device
    F = X.shape[-1]
    nbrs = model.policy.nbrs
    model.masked_steps = (0,)
    if saved:
        assert model.enable_saved_activations(), 'this policy cannot save its rollout activations'
    if compact:
        assert model.enable_compact_obs(), 'this policy cannot consume compact observations'

    def slab(x):                                  # [K,N,F] -> [K,N,n_obs]: own features, then the neighbours' (ascending)
        if compact:
            return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        s = np.zeros((K, N, model.policy.n_obs), dtype=np.float32)
        s[:, :, :F] = x
        for i in range(N):
            for k, j in enumerate(nbrs[i]):
                s[:, i, (k + 1) * F:(k + 2) * F] = x[:, j]
        return torch.from_numpy(s).to(dev)

    def uniforms(pi, a, live):                    # u [K,N] that makes the kernel draw action a from (about) pi
        cdf = np.cumsum(pi, axis=-1) / np.maximum(pi.sum(-1, keepdims=True), 1e-30)
        lo = np.where(a > 0, np.take_along_axis(cdf, np.maximum(a - 1, 0)[..., None], -1)[..., 0], 0.0)
        hi = np.take_along_axis(cdf, a[..., None], -1)[..., 0]
        u = np.where(live[:, None], 0.5 * (lo + hi), 0.5)
        return torch.from_numpy(u.astype(np.float32)).to(dev)

    zero = torch.zeros(K, dtype=torch.float32, device=dev)
    scratch = torch.zeros(K, N, dtype=torch.uint8, device=dev)
    PI = np.zeros_like(PIg)
    V = np.zeros_like(z['V'])
    model.reset_states()
    done_pre = torch.ones(K, dtype=torch.float32, device=dev)
    live = {0: np.arange(K) >= 2, 1: np.ones(K, bool)}      # replicas 0, 1 have no golden prefix (restarted afterwards)
    for ph in (0, 1):
        model.t = 0
        for t in range(T):
            model.buf_x[t].copy_(slab(X[:, ph, t]))
            d = done_pre if t == 0 else zero
            model.buf_done_pre[t].copy_(d)
            model.act(d, mode=ops.SAMPLE_UNIFORM, u=uniforms(PIg[:, ph, t], ACT[:, ph, t], live[ph]), done_is_zero=(t > 0))
            got = model.buf_act[t].cpu().numpy()
            assert np.array_equal(got[live[ph]], ACT[:, ph, t][live[ph]]), 'forced action draw failed at step %d' % t
            PI[:, ph, t] = model.buf_fp[t + 1].permute(1, 0, 2).cpu().numpy()
            if not saved:
                V[:, ph, t] = model.buf_v[t].t().cpu().numpy()
            model.t = t + 1
        model.buf_x[T].copy_(slab(X[:, ph, T]))
        v = model.bootstrap(zero, scratch, mode=ops.SAMPLE_UNIFORM, u=uniforms(PIg[:, ph, T], ACT[:, ph, T], live[ph]),
                            done_is_zero=True)
        PI[:, ph, T] = model._pi_boot.permute(1, 0, 2).cpu().numpy()
        V[:, ph, T] = v.t().cpu().numpy()
        if ph == 0:
            # batch boundary WITHOUT an update: what update() does to the rollout state (models.py: states_bw <-
            # states_fw, slot T becomes slot 0), then replicas 0, 1 start a new episode
            model.h_bw.copy_(model.h_fw)
            model.c_bw.copy_(model.c_fw)
            model.buf_fp[0].copy_(model.buf_fp[T])
            model.t = 0
            restart = torch.tensor([1, 1] + [0] * (K - 2), dtype=torch.uint8, device=dev)
            model.reset_states(mask=restart)
            done_pre = restart.to(torch.float32)
    last_done = torch.tensor([0, 1] + [0] * (K - 2), dtype=torch.uint8, device=dev)
    model.buf_done_post.zero_()
    model.buf_done_post[T - 1].copy_(last_done)
    model.load_rewards(torch.from_numpy(REW[:, 1].T.astype(np.float32)).to(dev).contiguous())
    R_end = (v * (1.0 - last_done.to(torch.float32)).view(1, -1)).contiguous()
    states = np.concatenate([model.c_fw.permute(1, 0, 2).cpu().numpy(), model.h_fw.permute(1, 0, 2).cpu().numpy()], axis=2)
    model.update(R_end)
    if saved:       # the values are complete (critic's neighbour-action term added) only after update()
        V[:, 1, :T] = model.buf_v.permute(2, 0, 1).cpu().numpy()
    tot = model.last_loss[3].cpu().numpy().astype(np.float64)
    gn = model.grad_norm.cpu().numpy().astype(np.float64)
    loss, gnorm = (tot, gn) if model.per_agent_optimizer else (np.array([tot.sum()]), gn[:1])
    return dict(PI=PI, V=V, RB=R_end.t().cpu().numpy(), STATES=states, LOSS=loss, GN=gnorm, saved=saved,
                STATS=var_stats_from_named(model.policy.params.ref_variables()))


def compare_batched(out, z, rtol_fw=1e-4, rtol_w=1e-3):
    """SURVEY.md 8(c) tolerances: forward rtol 1e-4, post-update weights rtol 1e-3."""
    K = int(z['K'])
    m = np.arange(K) >= 2
    # replica 1 ends its episode with the batch: the reference makes no bootstrap call for it (slot T stays 0)
    vm = np.ones_like(z['V'][:, 1], dtype=bool)
    vm[1, -1] = False
    np.testing.assert_allclose(out['PI'][:, 1][vm], z['PI'][:, 1][vm], rtol=rtol_fw, atol=1e-6, err_msg='pi (main batch)')
    np.testing.assert_allclose(out['PI'][m, 0], z['PI'][m, 0], rtol=rtol_fw, atol=1e-6, err_msg='pi (prefix)')
    if out.get('saved'):
        vm[:, -1] = False                        # bootstrap values are checked through RB; prefix values are not kept
    np.testing.assert_allclose(out['V'][:, 1][vm], z['V'][:, 1][vm], rtol=rtol_fw, atol=2e-5, err_msg='v')
    np.testing.assert_allclose(out['RB'], z['RB'], rtol=rtol_fw, atol=2e-5, err_msg='R bootstrap')
    keep = np.arange(K) != 1                      # replica 1: the product's (discarded) bootstrap call advanced its state
    np.testing.assert_allclose(out['STATES'][keep], z['STATES'][keep], rtol=rtol_fw, atol=2e-6, err_msg='states_fw')
    np.testing.assert_allclose(out['LOSS'], z['LOSS'], rtol=1e-4, atol=1e-5, err_msg='loss = mean of the replica losses')
    np.testing.assert_allclose(out['GN'], z['GN'], rtol=1e-4, atol=1e-6, err_msg='norm of the mean gradient')
    s, g = out['STATS'], z['STATS']
    np.testing.assert_allclose(s[..., 1:3], g[..., 1:3], rtol=rtol_w, err_msg='|w| / l2 after the update')
    np.testing.assert_allclose(s[..., 3:], g[..., 3:], rtol=rtol_w, atol=2e-6, err_msg='weight samples after the update')
