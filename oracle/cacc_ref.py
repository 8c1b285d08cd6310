"""NumPy restatement of the reference CACC platoon environment, batched over E
independent replicas.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/envs/cacc_env.py line by line:
  * config keys            cacc_env.py:320-343 (_load_config)
  * masks / n_s / a_map    cacc_env.py:253-283 (_init_space)
  * initial conditions     cacc_env.py:285-318 (_init_catchup/_init_slowdown), 166-189 (reset)
  * OVM controller         cacc_env.py:360-385 (get_vh/get_accel), 31-38 (_get_accel)
  * speed constraint       cacc_env.py:24-29  (_constrain_speed)
  * headway trapezoid      cacc_env.py:211-220
  * reward / collision     cacc_env.py:40-52  (_get_reward), 193-194, 229-237
  * observation            cacc_env.py:54-79  (_get_veh_state/_get_state)

Operation order inside every expression is kept identical to the reference so
that the float64 instance of this class reproduces the reference *bit for bit*
(pinned by tests/test_oracle_cacc.py against tests/golden/cacc_*.npz).
`dtype=np.float32` gives the fp32-cast variant used for tight kernel checks.
"""
import numpy as np

COLLISION_WT = 5          # cacc_env.py:9
COLLISION_HEADWAY = 10    # cacc_env.py:10
VDIFF = 5                 # cacc_env.py:11
A_MAP = ((0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (0.5, 0.5))  # cacc_env.py:275
DECEL_STEPS = 300         # cacc_env.py:317


class CaccParams:
    """The scalars of cacc_env.py:320-343, taken from an ini ENV_CONFIG section
    (configparser SectionProxy) or from keyword overrides."""

    def __init__(self, config=None, **kw):
        def g(k, d):
            if k in kw:
                return kw[k]
            if config is not None and k in config:
                return config.get(k)
            return d
        self.dt = float(g('control_interval_sec', 0.1))
        self.T = int(int(g('episode_length_sec', 60)) / self.dt)
        self.batch_size = int(g('batch_size', 60))
        self.h_min = float(g('headway_min', 1.0))
        self.h_star = float(g('headway_target', 20.0))
        self.h_s = float(g('headway_st', 5.0))
        self.h_g = float(g('headway_go', 35.0))
        self.v_max = float(g('speed_max', 30.0))
        self.v_star = float(g('speed_target', 15.0))
        self.u_min = float(g('accel_min', -2.5))
        self.u_max = float(g('accel_max', 2.5))
        self.name = g('scenario', 'cacc_catchup').split('_')[1]
        self.a = float(g('reward_v', 1.0))
        self.b = float(g('reward_u', 0.1))
        self.G = float(g('collision_penalty', 1000.0))
        self.n_agent = int(g('n_vehicle', 8))
        self.agent = g('agent', 'ma2c_nc')
        self.coop_gamma = float(g('coop_gamma', -1.0))
        self.seed = int(g('seed', 12))


def line_graph_masks(n):
    """cacc_env.py:253-268."""
    nb = np.zeros((n, n), dtype=int)
    dist = np.zeros((n, n), dtype=int)
    for i in range(n):
        for j in range(n):
            dist[i, j] = abs(i - j)
        if i >= 1:
            nb[i, i - 1] = 1
        if i <= n - 2:
            nb[i, i + 1] = 1
    return nb, dist


class CaccBatchRef:
    """E lock-stepped CACC replicas, float64 (or float32) NumPy."""

    def __init__(self, params, E=1, dtype=np.float64, train_mode=True):
        self.p = params
        self.E = E
        self.N = params.n_agent
        self.dtype = dtype
        self.train_mode = train_mode
        self.neighbor_mask, self.distance_mask = line_graph_masks(self.N)

    # ---- reset: cacc_env.py:166-189 with the uniform U supplied by the caller
    def reset(self, U, mask=None):
        """U[E]: the np.random.rand() draw of cacc_env.py:294 / :314 for each
        replica.  `mask[E]` (bool) restricts the reset to a subset (auto-reset)."""
        p, f = self.p, self.dtype
        U = np.asarray(U, dtype=f).reshape(self.E)
        h = np.ones((self.E, self.N), dtype=f) * f(p.h_star)
        v = np.ones((self.E, self.N), dtype=f) * f(p.v_star)
        v0_init = np.full(self.E, p.v_star, dtype=f)
        if p.name.startswith('catchup'):
            h[:, 0] = f(p.h_star) * (f(1.5) + U)                    # :294
        elif p.name.startswith('slowdown'):
            v[:] = (f(p.v_star) * (f(1.5) + U))[:, None]            # :314
            v0_init = v[:, 0].copy()                                # :317
        if mask is None:
            self.h, self.v, self.v0_init = h, v, v0_init
            self.u = np.zeros((self.E, self.N), dtype=f)
            self.t = np.zeros(self.E, dtype=np.int64)
            self.collided = np.zeros(self.E, dtype=bool)
        else:
            m = np.asarray(mask, dtype=bool)
            self.h[m], self.v[m], self.v0_init[m] = h[m], v[m], v0_init[m]
            self.u[m] = 0
            self.t[m] = 0
            self.collided[m] = False
        return self.obs()

    def v0(self, t):
        """Leading-vehicle speed profile v0s[t]: cacc_env.py:299 (catch-up,
        constant) / :316-318 (slow-down: np.linspace(v_init, v*, 300) then v*).
        np.linspace evaluates start + i*step with step=(stop-start)/(num-1) and
        pins the last sample to `stop`."""
        p, f = self.p, self.dtype
        t = np.asarray(t)
        if not p.name.startswith('slowdown'):
            return np.full(t.shape, p.v_star, dtype=f)
        step = (f(p.v_star) - self.v0_init) / f(DECEL_STEPS - 1)
        ramp = t.astype(f) * step + self.v0_init
        return np.where(t >= DECEL_STEPS - 1, f(p.v_star), ramp).astype(f)

    def vh(self, h):
        """OVM optimal velocity, cacc_env.py:360-369."""
        p, f = self.p, self.dtype
        mid = f(p.v_max) / 2 * (1 - np.cos(f(np.pi) * (h - f(p.h_s)) / (f(p.h_g) - f(p.h_s))))
        return np.where(h <= p.h_s, f(0), np.where(h < p.h_g, mid, f(p.v_max))).astype(f)

    # ---- step: cacc_env.py:191-242
    def step(self, action):
        p, f = self.p, self.dtype
        action = np.asarray(action).reshape(self.E, self.N)
        amap = np.asarray(A_MAP, dtype=f)
        alpha, beta = amap[action, 0], amap[action, 1]
        frozen = self.collided.copy()                                # :193
        h, v = self.h, self.v
        v_lead = np.concatenate([self.v0(self.t)[:, None], v[:, :-1]], axis=1)      # :33-37
        u_raw = alpha * (self.vh(h) - v) + beta * (v_lead - v)                      # :385
        v_next = v + np.clip(u_raw, f(p.u_min), f(p.u_max)) * f(p.dt)               # :26
        v_next = np.clip(v_next, f(0), f(p.v_max))                                  # :27
        u_c = (v_next - v) / f(p.dt)                                                # :28
        v_lead_next = np.concatenate([self.v0(self.t + 1)[:, None], v_next[:, :-1]], axis=1)
        h_next = h + f(0.5) * f(p.dt) * (v_lead + v_lead_next - v - v_next)         # :220
        live = ~frozen
        self.h = np.where(live[:, None], h_next, h).astype(f)
        self.v = np.where(live[:, None], v_next, v).astype(f)
        self.u = np.where(live[:, None], u_c, self.u).astype(f)
        # reward: cacc_env.py:40-52
        hit = live & (self.h.min(axis=1) < p.h_min)
        self.collided = self.collided | hit
        r = -(self.h - f(p.h_star)) ** 2
        r = r + (-f(p.a) * (self.v - f(p.v_star)) ** 2)
        r = r + (-f(p.b) * self.u ** 2)
        if self.train_mode:
            r = r + (-f(COLLISION_WT) * np.minimum(self.h - f(COLLISION_HEADWAY), f(0)) ** 2)
        reward = np.where(self.collided[:, None], -f(p.G) * np.ones_like(r), r).astype(f)
        self.t = self.t + 1
        global_reward = reward.sum(axis=1)                                          # :229
        done = (self.collided & (self.t % p.batch_size == 0)) | (self.t == p.T)     # :231-235
        if p.coop_gamma < 0:
            reward_out = global_reward                                              # :236-237
        else:
            reward_out = reward
        return self.obs(), reward_out, done, global_reward

    # ---- observation: cacc_env.py:54-65, compact 5 features per vehicle
    def veh_state(self):
        p, f = self.p, self.dtype
        h, v = self.h, self.v
        v_lead = np.concatenate([self.v0(self.t)[:, None], v[:, :-1]], axis=1)
        v_state = (v - f(p.v_star)) / f(p.v_star)
        vdiff = np.clip((v_lead - v) / f(VDIFF), -2, 2)
        vhdiff = np.clip((self.vh(h) - v) / f(VDIFF), -2, 2)
        h_state = (h + (v_lead - v) * f(p.dt) - f(p.h_star)) / f(p.h_star)
        u_state = self.u / f(p.u_max)
        return np.stack([v_state, vdiff, vhdiff, h_state, u_state], axis=-1).astype(f)

    def obs(self):
        """[E, N, 5] compact observation (the 'ma2c*' form of cacc_env.py:67-79)."""
        return self.veh_state()

    def obs_gathered(self, fp=None):
        """[E, N, 15] = own 5 (+) lower-index neighbour 5 (+) higher-index
        neighbour 5, zero padded for the edge vehicles, i.e. the 'ia2c' form
        of cacc_env.py:70-73 (ascending neighbour index) in a fixed-width slab.
        With `fp` [E,N,A] also returns the gathered neighbour fingerprints
        [E, N, 2A] (cacc_env.py:74-77)."""
        x = self.veh_state()
        return gather_line(x), (None if fp is None else gather_line(fp, own=False))

    def ref_obs_list(self, agent, fp=None, e=0):
        """The reference's list-of-1D-arrays observation for replica `e`."""
        x = self.veh_state()[e]
        out = []
        for i in range(self.N):
            cur = [x[i]]
            nb = np.where(self.neighbor_mask[i] == 1)[0]
            if agent.startswith('ia2c'):
                cur += [x[j] for j in nb]
            if agent == 'ia2c_fp':
                cur += [fp[e][j] for j in nb]
            out.append(np.concatenate(cur))
        return out


def gather_line(x, own=True):
    """Line-graph neighbour gather with left-packed ascending-index neighbours
    and zero padding: slot k of agent i holds its k-th neighbour."""
    E, N, F = x.shape
    m = 2
    y = np.zeros((E, N, (m + (1 if own else 0)) * F), dtype=x.dtype)
    off = 0
    if own:
        y[:, :, :F] = x
        off = F
    for i in range(N):
        nb = [j for j in (i - 1, i + 1) if 0 <= j < N]
        for k, j in enumerate(nb):
            y[:, i, off + k * F: off + (k + 1) * F] = x[:, j]
    return y
