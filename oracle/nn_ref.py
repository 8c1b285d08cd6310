"""CPU restatement of the reference's networks, loss, optimiser, n-step buffers and
model classes for ONE replica -- per-agent Python loops, per-step unrolled LSTMs,
exactly the structure of the TF-1 graphs.  TEST INFRASTRUCTURE (oracle/__init__.py);
also the `cpu_baseline` "port" timed by bench.py.

Follows (file:line under /root/reference):
  ortho_init              agents/utils.py:10-23
  fc / lstm               agents/utils.py:65-73, 87-115
  lstm_comm (NeurComm)    agents/utils.py:118-217
  lstm_ic3 (CommNet)      agents/utils.py:344-417
  heads                   agents/policies.py:50-77
  LstmPolicy / FPPolicy   agents/policies.py:80-185
  NC / IC3 policies       agents/policies.py:188-336, 429-476
  loss                    agents/policies.py:20-30, 232-255
  clip + RMSProp          agents/policies.py:32-39 (TF-1.12 clip_by_global_norm, ApplyRMSProp)
  buffers                 agents/utils.py:722-912
  IA2C / IA2C_FP / MA2C_* agents/models.py:15-292

PINNED against tests/golden/nn_*.npz, i.e. against the reference's own model code run on
oracle/tf1_shim (tests/test_oracle_nn.py); TF kernel semantics (RMSProp slots, global-norm
clip, softmax) are restated from the TF-1.12 definitions and stay unpinned at that boundary.
Arithmetic: torch CPU, float32 like TF (float64 optional for error analysis).
"""
import numpy as np
import torch


def ortho_init(shape, scale=np.sqrt(2)):
    a = np.random.standard_normal(shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == tuple(shape) else v
    return (scale * q.reshape(shape)).astype(np.float32)


class Vars:
    """Ordered variable store: creation order == np.random draw order of the reference."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.v = {}

    def w(self, name, shape):
        self.v[name] = torch.tensor(ortho_init(shape), dtype=self.dtype, requires_grad=True)

    def b(self, name, n):
        self.v[name] = torch.zeros(n, dtype=self.dtype, requires_grad=True)

    def __getitem__(self, k):
        return self.v[k]

    def scope(self, prefix):
        return [(k, p) for k, p in self.v.items() if k.startswith(prefix)]


def lstm_cell(x, c, h, done, wx, wh, b):
    """agents/utils.py:102-113."""
    c = c * (1 - done)
    h = h * (1 - done)
    z = x @ wx + h @ wh + b
    i, f, o, u = z.split(z.shape[-1] // 4, dim=-1)
    i, f, o, u = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(u)
    c = f * c + i * u
    h = o * torch.tanh(c)
    return c, h


def softmax(x):
    e = torch.exp(x - x.max(dim=-1, keepdim=True).values)
    return e / e.sum(dim=-1, keepdim=True)


class TFRMSProp:
    """clip_by_global_norm + RMSPropOptimizer over a fixed variable list (policies.py:32-39)."""

    def __init__(self, named, decay, eps, max_norm):
        self.named = named
        self.decay, self.eps, self.max_norm = decay, eps, max_norm
        self.ms = [torch.ones_like(p.detach()) for _, p in named]

    def step(self, loss, lr):
        params = [p for _, p in self.named]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
        norm = torch.sqrt(sum((g * g).sum() for g in grads))
        if self.max_norm > 0:
            scale = self.max_norm * torch.minimum(1.0 / norm, torch.tensor(1.0 / self.max_norm, dtype=norm.dtype))
            grads = [g * scale for g in grads]
        with torch.no_grad():
            for p, g, ms in zip(params, grads, self.ms):
                ms += (g * g - ms) * (1 - self.decay)
                p -= lr * g / torch.sqrt(ms + self.eps)
        return float(norm)


class OnPolicyBufferRef:
    """agents/utils.py:722-816 (single agent) / 819-912 (multi agent, `multi=True`)."""

    def __init__(self, gamma, alpha, distance_mask, multi=False):
        self.gamma, self.alpha, self.multi = gamma, alpha, multi
        if alpha > 0:
            self.distance_mask = np.asarray(distance_mask)
            self.max_distance = np.max(self.distance_mask, axis=-1)
        self.reset()

    def reset(self, done=False):
        self.obs, self.acts, self.rs, self.vs, self.adds, self.dones = [], [], [], [], [], [done]

    def add_transition(self, ob, na, a, r, v, done):
        self.obs.append(ob); self.adds.append(na); self.acts.append(a)
        self.rs.append(r); self.vs.append(v); self.dones.append(done)

    def _scan(self, R, vs, dist=None, maxd=None):
        Rs, Advs = [], []
        for r, v, done in zip(self.rs[::-1], vs[::-1], self.dones[:0:-1]):
            if self.alpha < 0:
                R = r + self.gamma * R * (1. - done)
            else:
                R = self.gamma * R * (1. - done)
                for t in range(maxd + 1):
                    R += (self.alpha ** t) * np.sum(np.asarray(r)[dist == t])
            Rs.append(R)
            Advs.append(R - v)
        return Rs[::-1], Advs[::-1]

    def sample_transition(self, R):
        if not self.multi:
            d = (self.distance_mask, self.max_distance) if self.alpha > 0 else (None, None)
            Rs, Advs = self._scan(R, self.vs, *d)
        else:
            vs = np.array(self.vs)
            Rs, Advs = [], []
            for i in range(vs.shape[1]):
                d = (self.distance_mask[i], self.max_distance[i]) if self.alpha > 0 else (None, None)
                a, b = self._scan(R[i], vs[:, i], *d)
                Rs.append(a); Advs.append(b)
        out = (self.obs, self.adds, self.acts, np.array(self.dones[:-1], dtype=bool),
               np.array(Rs, dtype=np.float32), np.array(Advs, dtype=np.float32))
        self.reset(self.dones[-1])
        return out


class _ModelBase:
    def __init__(self, n_s_ls, n_a_ls, neighbor_mask, distance_mask, coop_gamma, cfg, dtype=torch.float32):
        self.dtype = dtype
        self.nb = np.asarray(neighbor_mask)
        self.N = len(self.nb)
        self.nbr = [np.where(self.nb[i] == 1)[0] for i in range(self.N)]
        self.n_s_ls, self.A = list(n_s_ls), n_a_ls[0]
        self.H = cfg.getint('num_lstm')
        self.n_fc = cfg.getint('num_fc')
        self.n_step = cfg.getint('batch_size')
        self.reward_norm, self.reward_clip = cfg.getfloat('reward_norm'), cfg.getfloat('reward_clip')
        self.v_coef, self.e_coef = cfg.getfloat('value_coef'), cfg.getfloat('entropy_coef')
        self.gamma, self.lr = cfg.getfloat('gamma'), cfg.getfloat('lr_init')
        self.rms = (cfg.getfloat('rmsp_alpha'), cfg.getfloat('rmsp_epsilon'), cfg.getfloat('max_grad_norm'))
        self.coop_gamma = coop_gamma
        self.dist = np.asarray(distance_mask)
        self.vars = Vars(dtype)

    def t(self, x):
        return torch.as_tensor(np.asarray(x, dtype=np.float64)).to(self.dtype)

    def _norm_reward(self, reward):
        if self.reward_norm > 0:
            reward = reward / self.reward_norm
        if self.reward_clip > 0:
            reward = np.clip(reward, -self.reward_clip, self.reward_clip)
        return reward

    def _head(self, prefix_pi, prefix_v, h, na_onehot):
        v = self.vars
        pi = softmax(h @ v[prefix_pi + '/w'] + v[prefix_pi + '/b'])
        hv = torch.cat([h, na_onehot], dim=1) if na_onehot is not None else h
        val = (hv @ v[prefix_v + '/w'] + v[prefix_v + '/b']).squeeze(-1)
        return pi, val

    def _onehot(self, na):
        """one_hot(na)[T,m,A] -> [T, m*A] (policies.py:66-68)."""
        na = torch.as_tensor(np.asarray(na)).long().reshape(-1, np.asarray(na).shape[-1])
        return torch.nn.functional.one_hot(na, self.A).to(self.dtype).reshape(na.shape[0], -1)

    def _a2c_loss(self, pi, v, acts, Rs, Advs):
        """policies.py:20-30."""
        A_sparse = torch.nn.functional.one_hot(torch.as_tensor(np.asarray(acts)).long(), self.A).to(self.dtype)
        log_pi = torch.log(torch.clamp(pi, 1e-10, 1.0))
        entropy = -(pi * log_pi).sum(-1)
        entropy_loss = -entropy.mean() * self.e_coef
        policy_loss = -((log_pi * A_sparse).sum(-1) * self.t(Advs)).mean()
        value_loss = ((self.t(Rs) - v) ** 2).mean() * 0.5 * self.v_coef
        return policy_loss + value_loss + entropy_loss


class IA2CRef(_ModelBase):
    """IA2C (models.py:15-158) with N independent LstmPolicy (policies.py:80-154)."""
    fp = False

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        v, H, nf, A = self.vars, self.H, self.n_fc, self.A
        for i in range(self.N):
            m = len(self.nbr[i])
            s = 'lstm_%d' % i
            if self.fp:
                v.w(s + '/fcs/w', (self.n_s_ls[i], nf)); v.b(s + '/fcs/b', nf)
                v.w(s + '/fcp/w', (A * m, nf)); v.b(s + '/fcp/b', nf)
                v.w(s + '/lstm/wx', (2 * nf, 4 * H))
            else:
                v.w(s + '/fc/w', (self.n_s_ls[i], nf)); v.b(s + '/fc/b', nf)
                v.w(s + '/lstm/wx', (nf, 4 * H))
            v.w(s + '/lstm/wh', (H, 4 * H)); v.b(s + '/lstm/b', 4 * H)
            v.w(s + '/pi/w', (H, A)); v.b(s + '/pi/b', A)
            v.w(s + '/v/w', (H + A * m, 1)); v.b(s + '/v/b', 1)
        self.opt = [TFRMSProp(v.scope('lstm_%d/' % i), *self.rms) for i in range(self.N)]
        alpha = coop = self.coop_gamma
        self.buf = [OnPolicyBufferRef(self.gamma, alpha, self.dist[i]) for i in range(self.N)]
        self.reset()
        del coop

    def reset(self):
        self.states_fw = [torch.zeros(2 * self.H, dtype=self.dtype) for _ in range(self.N)]
        self.states_bw = [torch.zeros(2 * self.H, dtype=self.dtype) for _ in range(self.N)]

    def _net(self, i, obs, dones, nas, states):
        """_build_net (policies.py:136-149 / 163-185) over T steps."""
        v, s, nf = self.vars, 'lstm_%d' % i, self.n_fc
        ob = self.t(obs)
        if self.fp:
            n_x = self.n_s_ls[i]
            hx = torch.relu(ob[:, :n_x] @ v[s + '/fcs/w'] + v[s + '/fcs/b'])
            hp = torch.relu(ob[:, n_x:] @ v[s + '/fcp/w'] + v[s + '/fcp/b'])
            x = torch.cat([hx, hp], dim=1)
        else:
            x = torch.relu(ob @ v[s + '/fc/w'] + v[s + '/fc/b'])
        c, h = states[:self.H].unsqueeze(0), states[self.H:].unsqueeze(0)
        hs = []
        for t in range(x.shape[0]):
            c, h = lstm_cell(x[t:t + 1], c, h, float(dones[t]), v[s + '/lstm/wx'], v[s + '/lstm/wh'], v[s + '/lstm/b'])
            hs.append(h)
        hs = torch.cat(hs, dim=0)
        pi, val = self._head(s + '/pi', s + '/v', hs, self._onehot(nas) if nas is not None else None)
        return pi, val, torch.cat([c, h], dim=1).squeeze(0)

    def forward(self, obs, done, nactions=None, out_type='p'):
        out = []
        with torch.no_grad():
            for i in range(self.N):
                na = None if nactions is None else [nactions[i]]
                if out_type == 'p':
                    dummy = [np.zeros(len(self.nbr[i]), dtype=int)]
                    pi, _, st = self._net(i, [obs[i]], [done], dummy, self.states_fw[i])
                    self.states_fw[i] = st
                    out.append(pi[0].numpy())
                else:
                    _, val, _ = self._net(i, [obs[i]], [done], na, self.states_fw[i])
                    out.append(val[0].numpy())
        return out

    def add_transition(self, ob, naction, action, reward, value, done):
        reward = self._norm_reward(reward)
        for i in range(self.N):
            self.buf[i].add_transition(ob[i], naction[i], action[i], reward, value[i], done)

    def backward(self, Rends, dt=0):
        self.last = []
        for i in range(self.N):
            obs, nas, acts, dones, Rs, Advs = self.buf[i].sample_transition(Rends[i])
            pi, v, _ = self._net(i, np.array(obs), dones, np.array(nas), self.states_bw[i])
            loss = self._a2c_loss(pi, v, acts, Rs, Advs)
            gn = self.opt[i].step(loss, self.lr)
            self.last.append((float(loss.detach()), gn))
            self.states_bw[i] = self.states_fw[i].clone()


class IA2CFPRef(IA2CRef):
    """IA2C_FP (models.py:161-188): n_s_ls stays the env's; the policy input appends A*m fingerprints."""
    fp = True


class MA2CNCRef(_ModelBase):
    """MA2C_NC (models.py:191-258) with NCMultiAgentPolicy + lstm_comm."""
    scope, kind = 'nc', 'nc'

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        v, H, A = self.vars, self.H, self.A
        F = self.n_s_ls[0]
        for i in range(self.N):
            m = len(self.nbr[i])
            s = '%s/lstm_%s_%d' % (self.scope, 'comm' if self.kind == 'nc' else 'ic3', i)
            if self.kind == 'nc':
                v.w(s + '/w_msg', (H * m, H)); v.b(s + '/b_msg', H)
                v.w(s + '/w_ob', (F * (m + 1), H)); v.b(s + '/b_ob', H)
                v.w(s + '/w_fp', (A * m, H)); v.b(s + '/b_fp', H)
                v.w(s + '/wx_hid', (3 * H, 4 * H))
            else:
                v.w(s + '/w_msg', (H, H)); v.b(s + '/b_msg', H)
                v.w(s + '/w_ob', (F * (m + 1), H)); v.b(s + '/b_ob', H)
                v.w(s + '/wx_hid', (H, 4 * H))
            v.w(s + '/wh_hid', (H, 4 * H)); v.b(s + '/b_hid', 4 * H)
        for i in range(self.N):
            m = len(self.nbr[i])
            v.w('%s/pi_%d/w' % (self.scope, i), (H, A)); v.b('%s/pi_%d/b' % (self.scope, i), A)
            v.w('%s/v_%d/w' % (self.scope, i), (H + A * m, 1)); v.b('%s/v_%d/b' % (self.scope, i), 1)
        self.opt = TFRMSProp(v.scope(self.scope), *self.rms)
        self.buf = OnPolicyBufferRef(self.gamma, self.coop_gamma, self.dist, multi=True)
        self.reset()

    def reset(self):
        self.states_fw = torch.zeros(self.N, 2 * self.H, dtype=self.dtype)
        self.states_bw = torch.zeros(self.N, 2 * self.H, dtype=self.dtype)

    def _net(self, obs, ps, acts, dones, states):
        """obs [N,T,F], ps [N,T,A], acts [N,T] or None, dones [T], states [N,2H] (policies.py:275-312)."""
        v, H = self.vars, self.H
        xs, pp = self.t(obs), self.t(ps)
        T = xs.shape[1]
        c, h = states[:, :H], states[:, H:]
        hs = []
        for t in range(T):
            done = float(dones[t])
            out_m = h                                                   # un-masked previous h (Q3)
            nc, nh = [], []
            for i in range(self.N):
                js = self.nbr[i]
                s = '%s/lstm_%s_%d' % (self.scope, 'comm' if self.kind == 'nc' else 'ic3', i)
                xi = torch.cat([xs[i, t]] + [xs[j, t] for j in js]).unsqueeze(0)
                if self.kind == 'nc':
                    mi = torch.cat([out_m[j] for j in js]).unsqueeze(0)
                    pi_ = torch.cat([pp[j, t] for j in js]).unsqueeze(0)
                    si = torch.cat([torch.relu(xi @ v[s + '/w_ob'] + v[s + '/b_ob']),
                                    torch.relu(pi_ @ v[s + '/w_fp'] + v[s + '/b_fp']),
                                    torch.relu(mi @ v[s + '/w_msg'] + v[s + '/b_msg'])], dim=1)
                else:
                    mi = torch.stack([out_m[j] for j in js]).mean(0, keepdim=True)
                    si = torch.tanh(xi @ v[s + '/w_ob'] + v[s + '/b_ob']) + mi @ v[s + '/w_msg'] + v[s + '/b_msg']
                ci, hi = lstm_cell(si, c[i:i + 1], h[i:i + 1], done, v[s + '/wx_hid'], v[s + '/wh_hid'], v[s + '/b_hid'])
                nc.append(ci); nh.append(hi)
            c, h = torch.cat(nc, 0), torch.cat(nh, 0)
            hs.append(h)
        hs = torch.stack(hs, dim=1)                                      # [N,T,H]
        pis, vals = [], []
        for i in range(self.N):
            na = None
            if acts is not None:
                na = self._onehot(np.asarray(acts)[self.nb[i] == 1].T)   # [T, m]
            else:
                na = torch.zeros(T, self.A * len(self.nbr[i]), dtype=self.dtype)
            pi, val = self._head('%s/pi_%d' % (self.scope, i), '%s/v_%d' % (self.scope, i), hs[i], na)
            pis.append(pi); vals.append(val)
        return torch.stack(pis), torch.stack(vals), torch.cat([c, h], dim=1)

    def forward(self, obs, done, ps, actions=None, out_type='p'):
        with torch.no_grad():
            ob = np.asarray(obs)[:, None, :]
            p = np.asarray(ps)[:, None, :]
            a = None if actions is None else np.asarray(actions)[:, None]
            pi, val, st = self._net(ob, p, a, [done], self.states_fw)
            if out_type == 'p':
                self.states_fw = st
                return pi[:, 0].numpy()
            return val[:, 0].numpy()

    def add_transition(self, ob, p, action, reward, value, done):
        self.buf.add_transition(np.array(ob), np.array(p), action, self._norm_reward(reward), value, done)

    def backward(self, Rends, dt=0):
        obs, ps, acts, dones, Rs, Advs = self.buf.sample_transition(Rends)
        obs = np.transpose(np.array(obs, dtype=np.float32), (1, 0, 2))
        ps = np.transpose(np.array(ps, dtype=np.float32), (1, 0, 2))
        acts = np.transpose(np.array(acts))
        pi, v, _ = self._net(obs, ps, acts, dones, self.states_bw)
        # policies.py:232-255: mean over steps, SUM over agents
        A_sparse = torch.nn.functional.one_hot(torch.as_tensor(acts).long(), self.A).to(self.dtype)
        log_pi = torch.log(torch.clamp(pi, 1e-10, 1.0))
        entropy = -(pi * log_pi).sum(-1)
        prob_pi = (log_pi * A_sparse).sum(-1)
        entropy_loss = -entropy.mean(-1).sum() * self.e_coef
        policy_loss = -(prob_pi * self.t(Advs)).mean(-1).sum()
        value_loss = ((self.t(Rs) - v) ** 2).mean(-1).sum() * 0.5 * self.v_coef
        loss = policy_loss + value_loss + entropy_loss
        gn = self.opt.step(loss, self.lr)
        self.last = [(float(loss.detach()), gn)]
        self.states_bw = self.states_fw.clone()


class MA2CIC3Ref(MA2CNCRef):
    """MA2C_IC3 / CommNet (models.py:278-292, policies.py:429-476, lstm_ic3)."""
    scope, kind = 'ic3', 'ic3'


class MA2CDIALRef(MA2CNCRef):
    """MA2C_DIAL (models.py:295-309, policies.py:479-525, lstm_dial agents/utils.py:515-599)."""
    scope, kind = 'dial', 'dial'

    def __init__(self, *a, **k):
        _ModelBase.__init__(self, *a, **k)
        v, H, A = self.vars, self.H, self.A
        F = self.n_s_ls[0]
        for i in range(self.N):
            m = len(self.nbr[i])
            s = 'dial/lstm_comm_%d' % i
            v.w(s + '/w_msg', (H * m, H)); v.b(s + '/b_msg', H)
            v.w(s + '/w_ob', (F * (m + 1), H)); v.b(s + '/b_ob', H)
            v.w(s + '/wx_hid', (H, 4 * H)); v.w(s + '/wh_hid', (H, 4 * H)); v.b(s + '/b_hid', 4 * H)
        for i in range(self.N):                 # created inside the first unrolled step (agents/utils.py:561-564)
            v.w('dial/mfc_%d/w' % i, (H, H)); v.b('dial/mfc_%d/b' % i, H)
        for i in range(self.N):
            m = len(self.nbr[i])
            v.w('dial/pi_%d/w' % i, (H, A)); v.b('dial/pi_%d/b' % i, A)
            v.w('dial/v_%d/w' % i, (H + A * m, 1)); v.b('dial/v_%d/b' % i, 1)
        self.opt = TFRMSProp(v.scope('dial'), *self.rms)
        self.buf = OnPolicyBufferRef(self.gamma, self.coop_gamma, self.dist, multi=True)
        self.reset()

    def _net(self, obs, ps, acts, dones, states):
        v, H = self.vars, self.H
        xs, pp = self.t(obs), self.t(ps)
        T = xs.shape[1]
        c, h = states[:, :H], states[:, H:]
        hs = []
        for t in range(T):
            done = float(dones[t])
            out_m = torch.cat([torch.relu(h[i:i + 1] @ v['dial/mfc_%d/w' % i] + v['dial/mfc_%d/b' % i])
                               for i in range(self.N)], 0)
            nc, nh = [], []
            for i in range(self.N):
                js = self.nbr[i]
                s = 'dial/lstm_comm_%d' % i
                mi = torch.cat([out_m[j] for j in js]).unsqueeze(0)
                ai = torch.nn.functional.one_hot(torch.argmax(pp[i, t]), H).to(self.dtype).unsqueeze(0)
                xi = torch.cat([xs[i, t]] + [xs[j, t] for j in js]).unsqueeze(0)
                si = torch.relu(xi @ v[s + '/w_ob'] + v[s + '/b_ob']) + torch.relu(mi @ v[s + '/w_msg'] + v[s + '/b_msg']) + ai
                ci, hi = lstm_cell(si, c[i:i + 1], h[i:i + 1], done, v[s + '/wx_hid'], v[s + '/wh_hid'], v[s + '/b_hid'])
                nc.append(ci); nh.append(hi)
            c, h = torch.cat(nc, 0), torch.cat(nh, 0)
            hs.append(h)
        return self._heads(torch.stack(hs, dim=1), acts, T, torch.cat([c, h], dim=1))

    def _heads(self, hs, acts, T, st):
        pis, vals = [], []
        for i in range(self.N):
            if acts is not None:
                na = self._onehot(np.asarray(acts)[self.nb[i] == 1].T)
            else:
                na = torch.zeros(T, self.A * len(self.nbr[i]), dtype=self.dtype)
            pi, val = self._head('%s/pi_%d' % (self.scope, i), '%s/v_%d%s' % (self.scope, i, self.v_suffix), hs[i], na)
            pis.append(pi); vals.append(val)
        return torch.stack(pis), torch.stack(vals), st

    v_suffix = ''


class MA2CCURef(MA2CDIALRef):
    """IA2C_CU / ConseNet (models.py:261-275, policies.py:339-426)."""
    scope, kind, v_suffix = 'cu', 'cu', 'a'

    def __init__(self, *a, **k):
        _ModelBase.__init__(self, *a, **k)
        v, H, A = self.vars, self.H, self.A
        F = self.n_s_ls[0]
        for i in range(self.N):
            m = len(self.nbr[i])
            v.w('cu/fc_%da/w' % i, (F, H)); v.b('cu/fc_%da/b' % i, H)
            v.w('cu/lstm_%da/wx' % i, (H, 4 * H)); v.w('cu/lstm_%da/wh' % i, (H, 4 * H)); v.b('cu/lstm_%da/b' % i, 4 * H)
            v.w('cu/pi_%d/w' % i, (H, A)); v.b('cu/pi_%d/b' % i, A)
            v.w('cu/v_%da/w' % i, (H + A * m, 1)); v.b('cu/v_%da/b' % i, 1)
        self.opt = TFRMSProp(v.scope('cu'), *self.rms)
        self.buf = OnPolicyBufferRef(self.gamma, self.coop_gamma, self.dist, multi=True)
        self.reset()

    def _net(self, obs, ps, acts, dones, states):
        v, H = self.vars, self.H
        xs = self.t(obs)
        T = xs.shape[1]
        hs, cs, hl = [], [], []
        for i in range(self.N):
            x = torch.relu(xs[i] @ v['cu/fc_%da/w' % i] + v['cu/fc_%da/b' % i])
            c, h = states[i:i + 1, :H], states[i:i + 1, H:]
            out = []
            for t in range(T):
                c, h = lstm_cell(x[t:t + 1], c, h, float(dones[t]), v['cu/lstm_%da/wx' % i], v['cu/lstm_%da/wh' % i],
                                 v['cu/lstm_%da/b' % i])
                out.append(h)
            hs.append(torch.cat(out, 0)); cs.append(c); hl.append(h)
        st = torch.cat([torch.cat(cs, 0), torch.cat(hl, 0)], dim=1)
        return self._heads(torch.stack(hs, 0), acts, T, st)

    def backward(self, Rends, dt=0):
        super().backward(Rends, dt)
        # _consensus_update (policies.py:357-364): simultaneous mean over {i} + neighbours of the lstm_%da variables
        with torch.no_grad():
            for key in ('wx', 'wh', 'b'):
                old = [self.vars['cu/lstm_%da/%s' % (i, key)].detach().clone() for i in range(self.N)]
                for i in range(self.N):
                    grp = [old[i]] + [old[j] for j in self.nbr[i]]
                    self.vars['cu/lstm_%da/%s' % (i, key)].copy_(torch.stack(grp, -1).mean(-1))


REF_MODELS = {'ia2c': IA2CRef, 'ia2c_fp': IA2CFPRef, 'ma2c_nc': MA2CNCRef, 'ma2c_ic3': MA2CIC3Ref,
              'ma2c_cu': MA2CCURef, 'ma2c_dial': MA2CDIALRef}
