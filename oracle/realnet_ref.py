"""Synthetic (SUMO-free) Monaco-like ATSC network with HETEROGENEOUS agents: specification + NumPy oracle.
TEST INFRASTRUCTURE.

PARITY UNPINNED for the dynamics: the reference steps this scenario through an external SUMO process on a net file
that is not part of the repository (envs/real_net_env.py:145-150, envs/atsc_env.py:342-362).  This file IS the
specification of the synthetic model; the HIP kernel csrc/realnet.hip is checked against it.  Taken from the
reference and pinned by tests/test_oracle_realnet.py:
  * the 28 signalised nodes, their (directed) neighbour lists and phase sets        real_net_env.py:21-69
  * node order = sorted names, neighbor_mask[i, j] = 1 iff j is LISTED by i, distance_mask by BFS over the
    listed neighbours, -1 where unreachable                                        real_net_env.py:152-195
  * n_a_i = number of phases of node i (2..6), n_s_i = number of signal links = length of its phase strings
    (state `wave`, one entry per controlled link)                                   atsc_env.py:310-325, 373
  * 2 s yellow on links switching G->r, then green; control interval 5 s; T = 720   atsc_env.py:181-240
  * observation wave = vehicles on the detector of each link / norm_wave (no clipping: clip_wave = -1);
    reward_i = - sum of halting vehicles on the node's links (objective `queue`); per-agent rewards with the
    spatial discount coop_gamma = 0.9 of the shipped configs                        atsc_env.py:383-462; config_*_net.ini
  * demand: 4 flow groups, 5-min piecewise-constant activity 1,2,4,4,4,4,2,1,0,0,0 (groups 0,1) and
    0,0,0,1,2,4,4,4,4,2,1 (groups 2,3) flows of `flow_rate` veh/h each               real_net_data/build_file.py:70-96

Synthetic dynamics (store-and-forward fluid queues on the link graph, one control step = 5 s).  Link k of node i is
one lane with a queue q and a `transit` buffer (vehicles reaching the queue next step).
  sources: with m = number of listed neighbours of i, link k is fed by neighbour number (k mod (m+1)) of i, or is an
      EXTERNAL entry when k mod (m+1) == m (all links of a node without neighbours are external).  External link
      (i,k) belongs to flow group (i + k) mod 4; a group's demand  flow_rate * activity_g(t)  veh/h is split evenly
      over the group's external links and scaled by xi_g ~ U[0.8,1.2) per replica (Philox, stream RESET).
  1. effective green of link k from (previous phase, new phase): green->green 5 s, red->green 3 s, green->red 1 s
     (yellow clearance), red->red 0; a permitted 'g' serves at half rate;  D(i,k) = min(q, SAT * g_eff).
  2. node j offers out_j = sum_k D(j,k), split evenly over the fan_j links it feeds; link (i,k) accepts
     acc = min(out_src / fan_src, max(Q_MAX - q - transit, 0)); a node feeding nothing discharges out of the network.
  3. served(j,k) = D(j,k) * (sum of accepted / out_j)  (spill-back scales all links of the feeder alike).
  4. q' = q - served + transit;  transit' = acc (fed links) or the external arrivals of the step.
  5. detector count c = min(q', 7);  wave = c / norm_wave;  reward_i = - sum_k c(i,k).
"""
import numpy as np

# name | phase-set key | listed neighbours            (real_net_env.py:21-49, re-typed as a table)
_NODE_TABLE = """
10026 6.0 9431 9561 cluster_9563_9597 9531
8794 4.0 cluster_8985_9609 9837 9058 cluster_9563_9597
8940 2.1 9007 9429
8996 2.2
9007 2.3 9309 8940
9058 4.0 cluster_8985_9609 8794 joinedS_0
9153 2.0 9643
9309 4.0 9466 9007 cluster_9043_9052
9413 2.3 9721 9837
9429 5.0 cluster_9043_9052 8940
9431 2.4 9721 9884 9561 10026
9433 2.5
9466 4.0 9309 joinedS_0
9480 2.3
9531 2.6 joinedS_1
9561 4.0 cluster_9389_9689 10026
9643 2.3 9153
9713 3.0 9721
9721 6.0 9431 9713 9413
9837 3.1 9413 8794 cluster_8985_9609
9884 2.7 9713 cluster_9389_9689
cluster_8751_9630 4.0
cluster_8985_9609 4.0 9837 8794 9058
cluster_9043_9052 4.1 cluster_9563_9597 10026 joinedS_1
cluster_9389_9689 4.0 cluster_8751_9630 9884 9561 8996
cluster_9563_9597 4.2 10026 8794 joinedS_0 cluster_9043_9052
joinedS_0 6.1 9058 cluster_9563_9597 9466
joinedS_1 3.2 9531 9429
"""
# phase-set key -> phase strings over the node's signal links (real_net_env.py:51-69)
_PHASE_TABLE = """
4.0 GGgrrrGGgrrr rrrGGgrrrGGg rrGrrrrrGrrr rrrrrGrrrrrG
4.1 GGgrrGGGrrr rrGrrrrrrrr rrrGgrrrGGg rrrrGrrrrrG
4.2 GGGGrrrrrrrr GGggrrGGggrr rrrGGGGrrrrr grrGGggrrGGg
2.0 GGrrr ggGGG
2.1 GGGrrr rrGGGg
2.2 Grr gGG
2.3 GGGgrr GrrrGG
2.4 GGGGrr rrrrGG
2.5 Gg rG
2.6 GGGg rrrG
2.7 GGg rrG
3.0 GGgrrrGGg rrGrrrrrG rrrGGGGrr
3.1 GgrrGG rGrrrr rrGGGr
3.2 GGGGrrrGG rrrrGGGGr GGGGrrGGr
5.0 GGGGgrrrrGGGggrrrr grrrGrrrrgrrGGrrrr GGGGGrrrrrrrrrrrrr rrrrrrrrrGGGGGrrrr rrrrrGGggrrrrrggGg
6.0 GGGgrrrGGGgrrr rrrGrrrrrrGrrr GGGGrrrrrrrrrr rrrrrrrrrrGGGG rrrrGGgrrrrGGg rrrrrrGrrrrrrG
6.1 GGgrrGGGrrrGGGgrrrGGGg rrGrrrrrrrrrrrGrrrrrrG GGGrrrrrGGgrrrrGGgrrrr GGGrrrrrrrGrrrrrrGrrrr rrrGGGrrrrrrrrrrrrGGGG rrrGGGrrrrrGGGgrrrGGGg
"""
DT, YELLOW, YELLOW_EFF = 5.0, 2.0, 1.0
SAT, Q_MAX, DET_CAP = 0.5, 26.0, 7.0
ACTIVITY = np.array([[1, 2, 4, 4, 4, 4, 2, 1, 0, 0, 0]] * 2 + [[0, 0, 0, 1, 2, 4, 4, 4, 4, 2, 1]] * 2, dtype=np.float64)
N_GROUP = 4


class Topology:
    """Everything static about the network, as plain arrays (shared by the oracle, the host env and the kernel)."""

    def __init__(self):
        rows = [ln.split() for ln in _NODE_TABLE.strip().splitlines()]
        phases = {ln.split()[0]: ln.split()[1:] for ln in _PHASE_TABLE.strip().splitlines()}
        nodes = {r[0]: (r[1], r[2:]) for r in rows}
        self.names = sorted(nodes)                                   # real_net_env.py:189
        N = self.N = len(self.names)
        idx = {n: i for i, n in enumerate(self.names)}
        self.phases = [phases[nodes[n][0]] for n in self.names]
        self.n_a_ls = [len(p) for p in self.phases]
        self.n_s_ls = [len(p[0]) for p in self.phases]
        assert all(len(s) == self.n_s_ls[i] for i, p in enumerate(self.phases) for s in p)
        self.A, self.L = max(self.n_a_ls), max(self.n_s_ls)
        self.nbrs_listed = [[idx[m] for m in nodes[n][1]] for n in self.names]       # in the order of the table
        self.neighbor_mask = np.zeros((N, N), dtype=int)
        for i, js in enumerate(self.nbrs_listed):
            self.neighbor_mask[i, js] = 1
        self.nbrs = [sorted(js) for js in self.nbrs_listed]          # ascending index: boolean_mask order of the nets
        self.m_max = max(len(js) for js in self.nbrs)
        self.distance_mask = -np.ones((N, N), dtype=int)             # BFS over the listed neighbours, -1 = unreachable
        for i in range(N):
            self.distance_mask[i, i] = 0
            frontier, d = [i], 0
            while frontier:
                d += 1
                nxt = []
                for u in frontier:
                    for v in self.nbrs_listed[u]:
                        if self.distance_mask[i, v] < 0:
                            self.distance_mask[i, v] = d
                            nxt.append(v)
                frontier = nxt
        # signal tables: green[i, a, k] = 0 r / 1 G / 2 g (padded with r)
        self.green = np.zeros((N, self.A, self.L), dtype=np.int32)
        for i, p in enumerate(self.phases):
            for a, s in enumerate(p):
                self.green[i, a, :len(s)] = [{'r': 0, 'G': 1, 'g': 2}[ch] for ch in s]
        # link sources (-1 = external) and the feeders' fan-out lists
        self.src = -np.ones((N, self.L), dtype=np.int32)
        for i in range(N):
            m = len(self.nbrs[i])
            for k in range(self.n_s_ls[i]):
                slot = k % (m + 1)
                if slot < m:
                    self.src[i, k] = self.nbrs[i][slot]
        self.fan = np.array([(self.src == j).sum() for j in range(N)], dtype=np.int32)
        self.group = -np.ones((N, self.L), dtype=np.int32)
        for i in range(N):
            for k in range(self.n_s_ls[i]):
                if self.src[i, k] < 0:
                    self.group[i, k] = (i + k) % N_GROUP
        n_ext = np.array([(self.group == g).sum() for g in range(N_GROUP)])
        assert n_ext.min() > 0
        # share of its group's demand that an external link receives
        self.ext_share = np.where(self.group >= 0, 1.0 / n_ext[np.maximum(self.group, 0)], 0.0)


TOPO = Topology()


def activity(group, sec):
    piece = int(sec) // 300
    return float(ACTIVITY[group, piece]) if piece < ACTIVITY.shape[1] else 0.0


class NetParams:
    def __init__(self, config=None, **kw):
        def g(k, d):
            if k in kw:
                return kw[k]
            if config is not None and k in config:
                return config.get(k)
            return d
        self.control = int(g('control_interval_sec', 5))
        self.yellow = int(g('yellow_interval_sec', 2))
        self.episode_sec = int(g('episode_length_sec', 3600))
        self.T = int(np.ceil(self.episode_sec / self.control))
        self.norm_wave = float(g('norm_wave', 1.0))
        self.clip_wave = float(g('clip_wave', -1))
        self.flow_rate = float(g('flow_rate', 325))
        self.coop_gamma = float(g('coop_gamma', 0.9))
        self.agent = g('agent', 'ma2c_nc')
        self.seed = int(g('seed', 12))
        assert self.control == 5 and self.yellow == 2, 'the synthetic model is specified for 5 s / 2 s'


class NetBatchRef:
    """E replicas of the synthetic network, float64 (or float32) NumPy."""

    def __init__(self, params, E=1, dtype=np.float64, topo=TOPO):
        self.p, self.E, self.f, self.tp = params, E, dtype, topo
        self.valid = (np.arange(topo.L)[None, :] < np.array(topo.n_s_ls)[:, None])        # [N,L]

    def reset(self, xi, mask=None):
        f, tp = self.f, self.tp
        xi = np.asarray(xi, dtype=f).reshape(self.E, N_GROUP)
        if mask is None:
            self.q = np.zeros((self.E, tp.N, tp.L), dtype=f)
            self.tr = np.zeros((self.E, tp.N, tp.L), dtype=f)
            self.prev = np.zeros((self.E, tp.N), dtype=np.int64)
            self.t = np.zeros(self.E, dtype=np.int64)
            self.xi = xi.copy()
        else:
            m = np.asarray(mask, dtype=bool)
            self.q[m] = 0; self.tr[m] = 0; self.prev[m] = 0; self.t[m] = 0
            self.xi[m] = xi[m]
        return self.obs()

    def _eff_green(self, prev, cur):
        tp = self.tp
        n = np.arange(tp.N)[None, :]
        gp = tp.green[n, prev] != 0                      # [E,N,L]
        code = tp.green[n, cur]
        gc = code != 0
        g = np.where(gc & gp, DT, np.where(gc & ~gp, DT - YELLOW, np.where(~gc & gp, YELLOW_EFF, 0.0)))
        same = (prev == cur)[..., None]
        g = np.where(same, np.where(gc, DT, 0.0), g)
        return (g * np.where(code == 2, 0.5, 1.0)).astype(self.f)

    def step(self, action):
        f, E, tp = self.f, self.E, self.tp
        a = np.asarray(action).reshape(E, tp.N).astype(np.int64)
        geff = self._eff_green(self.prev, a) * self.valid
        D = np.minimum(self.q, f(SAT) * geff)
        out = D.sum(axis=2)                                                            # [E,N]
        space = np.maximum(f(Q_MAX) - self.q - self.tr, f(0))
        fed = tp.src >= 0
        srcc = np.maximum(tp.src, 0)
        offer = np.where(fed[None], out[:, srcc] / np.maximum(tp.fan[srcc], 1).astype(f)[None], f(0))   # [E,N,L]
        acc = np.minimum(offer, space) * fed[None]
        delivered = np.zeros((E, tp.N), dtype=f)
        for i in range(tp.N):                                   # fixed order: ascending (node, link) of the fed links
            for k in range(tp.n_s_ls[i]):
                if fed[i, k]:
                    delivered[:, tp.src[i, k]] += acc[:, i, k]
        delivered = np.where(tp.fan[None] == 0, out, delivered)
        scale = np.where(out > f(1e-6), delivered / np.maximum(out, f(1e-6)), f(0))
        served = D * scale[:, :, None]
        sec = self.t * self.p.control
        act = np.array([[activity(g, s) for g in range(N_GROUP)] for s in sec], dtype=f)   # [E,4]
        grp = np.maximum(tp.group, 0)
        gfac = (f(self.p.flow_rate) * act / f(3600) * f(DT) * self.xi).astype(f)       # [E,4] arrivals of each flow group
        ext = gfac[:, grp] * tp.ext_share[None].astype(f) * (tp.group >= 0)[None]
        self.q = (self.q - served + self.tr).astype(f)
        self.tr = (acc + ext).astype(f)
        self.prev = a
        self.t = self.t + 1
        c = np.minimum(self.q, f(DET_CAP)) * self.valid
        reward = -c.sum(axis=2)
        g = reward.sum(axis=1)
        done = self.t >= self.p.T
        r_out = g if self.p.coop_gamma < 0 else reward
        return self.obs(), r_out.astype(f), done, g.astype(f)

    def obs(self):
        f = self.f
        c = np.minimum(self.q, f(DET_CAP)) / f(self.p.norm_wave)
        if self.p.clip_wave >= 0:
            c = np.clip(c, 0, f(self.p.clip_wave))
        return (c * self.valid).astype(f)


def gather_net(x, topo=TOPO):
    """[E,N,L] -> [E,N,L*(1+m_max)]: own features then the neighbours' in ascending node index, one L-wide slot each
    (zero padded: the padded layout of the heterogeneous nets, deeprl_network_amd/agents/policies.py)."""
    E, N, L = x.shape
    y = np.zeros((E, N, L * (1 + topo.m_max)), dtype=x.dtype)
    y[:, :, :L] = x
    for i in range(N):
        for k, j in enumerate(topo.nbrs[i]):
            y[:, i, (k + 1) * L:(k + 2) * L] = x[:, j]
    return y
