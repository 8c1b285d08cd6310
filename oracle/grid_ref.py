"""Synthetic (SUMO-free) 5x5 ATSC grid: specification + NumPy oracle.  TEST INFRASTRUCTURE.

PARITY UNPINNED: the reference steps this scenario through an external SUMO process
(envs/atsc_env.py:342-362) that is not available, so there is nothing to pin the
*dynamics* against.  This file IS the specification of the synthetic model; the HIP
kernel csrc/grid.hip is checked against it.  What is taken from the reference and is
pinned by tests/test_oracle_grid.py:
  * topology, neighbour mask, hop-distance mask      large_grid_env.py:58-105
  * 5 phases over 12 signal links                    large_grid_env.py:23-27
  * 2 s yellow on links switching G->r, then green   atsc_env.py:181-186, 216-240
  * observation `wave` = vehicles on the 50 m detector of each of the 12 controlled
    links / norm_wave, clipped to clip_wave           atsc_env.py:420-462, 502-504
  * reward = - sum of halting vehicles on the 12 links (objective `queue`),
    global sum when coop_gamma < 0                    atsc_env.py:383-418, 205-206
  * episode 3600 s, control interval 5 s (T = 720)    atsc_env.py:81-84, 189-191
  * demand: 4 flow groups x 3 entries, 5-min piecewise-constant veh/h from
    peak_flow1 / peak_flow2 and the two ratio profiles, second wave from 900 s
                                                       large_grid_data/build_file.py:268-326
  * n_s = 12 per node with the duplicated lanes of SURVEY.md 8a (approach order
    N,E,S,W; links right/straight/left; avenues 1 lane, streets 2 lanes)

Synthetic dynamics (store-and-forward fluid queues, one control step = 5 s):
  node i = row*5 + col (nt{i+1} of the reference), row 0 at the bottom; neighbours N=i+5,
  E=i+1, S=i-5, W=i-1.  Each node has 6 physical incoming lanes
      0: N approach (avenue)   1: E approach lane 0 (right+straight)   2: E lane 1 (left)
      3: S approach (avenue)   4: W approach lane 0                    5: W lane 1
  state per lane: queue q (veh) and `transit` (veh that arrive at the queue next step).
  1. effective green per link k from (previous phase, new phase):
        green->green 5 s, red->green 3 s (red during the 2 s yellow), green->red 1 s
        (yellow clearance), red->red 0; permitted left 'g' serves at half rate.
  2. desired link flow  D_k = min(q_lane(k) * share_k, SAT * g_k * share_k * f_k)
        shares: avenue lane r/s/l = .2/.6/.2; street lane 0 r/s = .15/.85, .7/.85;
        street lane 1 is left only.
  3. every link feeds one approach of one neighbour (or leaves the grid); the receiving
     approach offers space = sum_lanes max(Q_MAX - q - transit, 0) and all feeders are
     scaled by min(1, space / sum D)  (spill-back).
  4. q' = q - served + transit;  transit' = received inflow (streets: 85 % lane 0, 15 %
     lane 1) + external arrivals  rate(t)/3600 * 5 s * xi_group,  xi ~ U[0.8, 1.2) per
     replica and flow group (Philox, stream RESET).
  5. detector count c = min(q, 7) (50 m / 7.5 m); wave_k = min(c / norm_wave, clip_wave);
     reward_i = - sum_k c_lane(k)  (12 links, duplicated lanes counted like the reference).
  6. objectives `wait` / `hybrid` (atsc_env.py:383-418: the waiting time of the FRONT vehicle of every detector,
     `getWaitingTime` of the car with the largest lane position): a fluid queue has no vehicles, so the spec keeps one more
     state per lane, head_wait (s): the lane's front vehicle has been standing since the lane last discharged,
         head_wait' = 0 if the lane served more than WAIT_EPS = 1e-3 veh this step or held at most WAIT_EPS at its start,
         else head_wait + 5 s;
     wait_i = sum_k head_wait'_lane(k) over the 12 links (duplicated lanes like the queue count);
     reward_i = - wait_i (`wait`)  or  - queue_i - coef_wait * wait_i (`hybrid`).  Reset clears it.  (Outside every shipped
     config: all of them set objective = queue.)
"""
import numpy as np

N_NODE, SIDE, N_LINK, N_LANE, N_PHASE = 25, 5, 12, 6, 5
PHASES = ['GGgrrrGGgrrr', 'rrrGrGrrrGrG', 'rrrGGrrrrGGr', 'rrrGGGrrrrrr', 'rrrrrrrrrGGG']   # large_grid_env.py:25-26
DT, YELLOW = 5.0, 2.0
SAT = 0.5            # veh/s per lane
Q_MAX = 26.0         # 200 m block / 7.5 m
DET_CAP = 7.0        # 50 m detector / 7.5 m
YELLOW_EFF = 1.0
WAIT_EPS = 1e-3      # veh: a standing queue / a discharge below this does not count (step 6)
LINK_LANE = np.array([0, 0, 0, 1, 1, 2, 3, 3, 3, 4, 4, 5])
LINK_SHARE = np.array([.2, .6, .2, .15 / .85, .7 / .85, 1.0] * 2)
LANE_APPROACH = np.array([0, 1, 1, 2, 3, 3])
APPROACH_SPLIT = np.array([1.0, 0.85, 0.15, 1.0, 0.85, 0.15])        # inflow share of each lane of its approach
# link k of a node -> (neighbour offset in (drow, dcol), receiving approach)
LINK_DEST = [(0, -1, 1), (-1, 0, 0), (0, 1, 3),      # from N (heading south): right->west nbr's E, straight, left
             (1, 0, 2), (0, -1, 1), (-1, 0, 0),      # from E (heading west)
             (0, 1, 3), (1, 0, 2), (0, -1, 1),       # from S (heading north)
             (-1, 0, 0), (0, 1, 3), (1, 0, 2)]       # from W (heading east)
# approach r of a node is fed by these 3 links of the neighbour in direction APPROACH_FROM[r]
APPROACH_FROM = [(1, 0), (0, 1), (-1, 0), (0, -1)]
APPROACH_FEED = [(1, 5, 9), (0, 4, 8), (3, 7, 11), (2, 6, 10)]
RATIOS1 = np.array([0.4, 0.7, 0.9, 1.0, 0.75, 0.5, 0.25])      # build_file.py:298
RATIOS2 = np.array([0.3, 0.8, 0.9, 1.0, 0.8, 0.6, 0.2])        # build_file.py:299
# 12 external entries: (node index 0-based, approach, flow group)   build_file.py:285-295 + edge_maps
ENTRIES = [(23, 0, 0), (22, 0, 0), (21, 0, 0),      # np12,13,14 -> nt24,23,22 from the north
           (20, 3, 1), (10, 3, 1), (0, 3, 1),       # np16,18,20 -> nt21,11,1 from the west
           (1, 2, 2), (2, 2, 2), (3, 2, 2),         # np2,3,4 -> nt2,3,4 from the south
           (4, 1, 3), (14, 1, 3), (24, 1, 3)]       # np6,8,10 -> nt5,15,25 from the east


def grid_masks():
    """neighbor_mask / distance_mask of large_grid_env.py:58-105."""
    nb = np.zeros((N_NODE, N_NODE), dtype=int)
    dist = np.zeros((N_NODE, N_NODE), dtype=int)
    for i in range(N_NODE):
        for j in range(N_NODE):
            dist[i, j] = abs(i // SIDE - j // SIDE) + abs(i % SIDE - j % SIDE)
    nb[dist == 1] = 1
    return nb, dist


def green_table():
    """[5,12] 0 = r, 1 = G, 2 = g."""
    return np.array([[{'r': 0, 'G': 1, 'g': 2}[c] for c in p] for p in PHASES])


def demand_rate(group, sec, peak1, peak2):
    """veh/h of one entry of flow group `group` at time `sec` (build_file.py:296-321)."""
    piece = int(sec) // 300
    if group in (0, 1):
        if piece >= 7:
            return 0.0
        return peak1 * (0.6 if group == 0 else 1.0) * RATIOS1[piece]
    if piece < 3 or piece >= 10:
        return 0.0
    return peak2 * (0.6 if group == 2 else 1.0) * RATIOS2[piece - 3]


class GridParams:
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
        self.norm_wave = float(g('norm_wave', 5.0))
        self.clip_wave = float(g('clip_wave', 2.0))
        self.peak1 = float(g('peak_flow1', 1100))
        self.peak2 = float(g('peak_flow2', 925))
        self.coop_gamma = float(g('coop_gamma', -1))
        self.agent = g('agent', 'ma2c_ic3')
        self.seed = int(g('seed', 12))
        self.objective = str(g('objective', 'queue'))            # atsc_env.py:87
        self.coef_wait = float(g('coef_wait', 0.0))              # atsc_env.py:96
        assert self.objective in ('queue', 'wait', 'hybrid')
        assert self.control == 5 and self.yellow == 2, 'the synthetic model is specified for 5 s / 2 s'


class GridBatchRef:
    def __init__(self, params, E=1, dtype=np.float64):
        self.p, self.E, self.f = params, E, dtype
        self.green = green_table()
        self.nb, self.dist = grid_masks()

    def reset(self, xi, mask=None):
        """xi [E,4] demand scale of each flow group (0.8 + 0.4 U)."""
        f = self.f
        xi = np.asarray(xi, dtype=f).reshape(self.E, 4)
        if mask is None:
            self.q = np.zeros((self.E, N_NODE, N_LANE), dtype=f)
            self.tr = np.zeros((self.E, N_NODE, N_LANE), dtype=f)
            self.prev = np.zeros((self.E, N_NODE), dtype=np.int64)       # atsc_env.py:509-513
            self.t = np.zeros(self.E, dtype=np.int64)
            self.xi = xi.copy()
            self.hw = np.zeros((self.E, N_NODE, N_LANE), dtype=f)
        else:
            m = np.asarray(mask, dtype=bool)
            self.q[m] = 0; self.tr[m] = 0; self.prev[m] = 0; self.t[m] = 0; self.hw[m] = 0
            self.xi[m] = xi[m]
        return self.obs()

    def _eff_green(self, prev, cur):
        gp = self.green[prev] != 0          # [E,N,12]
        gc = self.green[cur] != 0
        same = (prev == cur)[..., None]
        g = np.where(gc & gp, DT, np.where(gc & ~gp, DT - YELLOW, np.where(~gc & gp, YELLOW_EFF, 0.0)))
        g = np.where(same, np.where(gc, DT, 0.0), g)
        fac = np.where(self.green[cur] == 2, 0.5, 1.0)
        return (g * fac).astype(self.f)

    def step(self, action):
        f, E = self.f, self.E
        a = np.asarray(action).reshape(E, N_NODE).astype(np.int64)
        geff = self._eff_green(self.prev, a)                                        # [E,N,12] (incl. 'g' factor)
        share = LINK_SHARE.astype(f)
        qk = self.q[:, :, LINK_LANE]                                                # [E,N,12]
        D = np.minimum(qk * share, f(SAT) * geff * share)
        space = np.zeros((E, N_NODE, 4), dtype=f)
        free = np.maximum(f(Q_MAX) - self.q - self.tr, f(0))
        for lane in range(N_LANE):
            space[:, :, LANE_APPROACH[lane]] += free[:, :, lane]
        insum = np.zeros((E, N_NODE, 4), dtype=f)
        for n in range(N_NODE):
            r0, c0 = divmod(n, SIDE)
            for ap in range(4):
                dr, dc = APPROACH_FROM[ap]
                rr, cc = r0 + dr, c0 + dc
                if 0 <= rr < SIDE and 0 <= cc < SIDE:
                    m = rr * SIDE + cc
                    k0, k1, k2 = APPROACH_FEED[ap]
                    insum[:, n, ap] = D[:, m, k0] + D[:, m, k1] + D[:, m, k2]
        scale = np.minimum(f(1), space / np.maximum(insum, f(1e-6)))
        flow = np.zeros_like(D)
        for n in range(N_NODE):
            r0, c0 = divmod(n, SIDE)
            for k in range(N_LINK):
                dr, dc, ap = LINK_DEST[k]
                rr, cc = r0 + dr, c0 + dc
                if 0 <= rr < SIDE and 0 <= cc < SIDE:
                    flow[:, n, k] = D[:, n, k] * scale[:, rr * SIDE + cc, ap]
                else:
                    flow[:, n, k] = D[:, n, k]                                      # leaves the grid
        served = np.zeros_like(self.q)
        for k in range(N_LINK):
            served[:, :, LINK_LANE[k]] += flow[:, :, k]
        inflow = insum * scale                                                      # [E,N,4]
        sec = self.t * self.p.control
        for (node, ap, grp) in ENTRIES:
            rate = np.array([demand_rate(grp, s, self.p.peak1, self.p.peak2) for s in sec], dtype=f)
            inflow[:, node, ap] += rate / f(3600) * f(DT) * self.xi[:, grp]
        split = APPROACH_SPLIT.astype(f)
        moved = (served > f(WAIT_EPS)) | (self.q <= f(WAIT_EPS))                                        # step 6: the front vehicle left / no queue stood
        self.hw = np.where(moved, f(0), self.hw + f(DT)).astype(f)
        self.q = (self.q - served + self.tr).astype(f)
        self.tr = (inflow[:, :, LANE_APPROACH] * split).astype(f)
        self.prev = a
        self.t = self.t + 1
        c = np.minimum(self.q, f(DET_CAP))[:, :, LINK_LANE]                         # [E,N,12]
        reward = -c.sum(axis=2)
        if self.p.objective != 'queue':
            wait = self.hw[:, :, LINK_LANE].sum(axis=2)
            reward = -wait if self.p.objective == 'wait' else reward - f(self.p.coef_wait) * wait
        g = reward.sum(axis=1)
        done = self.t >= self.p.T
        r_out = g if self.p.coop_gamma < 0 else reward
        return self.obs(), r_out.astype(f), done, g.astype(f)

    def obs(self):
        f = self.f
        c = np.minimum(self.q, f(DET_CAP))[:, :, LINK_LANE] / f(self.p.norm_wave)
        if self.p.clip_wave >= 0:
            c = np.clip(c, 0, f(self.p.clip_wave))
        return c.astype(f)


def gather_grid(x, nb=None):
    """[E,25,F] -> [E,25,5F]: own features then the neighbours' in ascending node index
    (tf.boolean_mask order of lstm_ic3 / lstm_comm), left packed, zero padded."""
    if nb is None:
        nb, _ = grid_masks()
    E, N, F = x.shape
    y = np.zeros((E, N, 5 * F), dtype=x.dtype)
    y[:, :, :F] = x
    for i in range(N):
        for k, j in enumerate(np.where(nb[i] == 1)[0]):
            y[:, i, (k + 1) * F:(k + 2) * F] = x[:, j]
    return y
