"""A minimal fake `tensorflow` (TF-1.12 graph-mode API surface) executed with
torch on the CPU.  TEST INFRASTRUCTURE ONLY -- never imported by the product.

Purpose: TensorFlow 1.12 is not installable here (no network), so the
reference's own graph-building code (/root/reference/agents/utils.py,
agents/policies.py, agents/models.py, utils.py) cannot be executed as is.
With this package first on sys.path, `import tensorflow as tf` inside the
UNMODIFIED reference resolves to this shim and the reference's Python builds
its graphs out of lazy nodes that are evaluated with torch.  That pins every
*structural* decision of the restatement (which weight multiplies what, concat
/ mask / neighbour order, variable creation order of the np.random draws, done
masking, loss reductions) against the reference itself
(tests/golden/make_golden_nn.py -> tests/golden/nn_*.npz).

What is NOT executed but restated here, from the TF-1.12 kernel definitions:
  * RMSPropOptimizer (ApplyRMSProp): slots ms=1, mom=0;
        ms  += (g*g - ms) * (1 - decay)
        mom  = mom * momentum + lr * g / sqrt(ms + epsilon)
        var -= mom
  * clip_by_global_norm: norm = sqrt(sum ||t||^2);
        scale = clip * min(1/norm, 1/clip);  t_i *= scale
  * softmax = exp(x - max) / sum, one_hot, boolean_mask (ascending index).
Those remain "parity unpinned at the TF kernel boundary" (DESIGN.md).

Design: every op creates a Node holding a python closure over torch; the node
is evaluated once at build time on zero placeholders to learn its static shape
(`x.shape[1].value` in the reference needs it), and again per Session.run with
the fed values.  Evaluation is iterative in creation order (graphs here have
~10^4 nodes and chains deeper than the recursion limit).
"""
import os

import numpy as np
import torch

float32 = 'float32'
int32 = 'int32'
bool = 'bool'  # noqa: A001

# compute dtype of the shim graph: float64 (default) makes the golden vectors a
# rounding-free evaluation of the reference graph; NMARL_SHIM_DTYPE=float32 mimics TF.
_FDT = torch.float64 if os.environ.get('NMARL_SHIM_DTYPE', 'float64') == 'float64' else torch.float32


class Dim(int):
    @property
    def value(self):
        return int(self)


class _Graph:
    def __init__(self):
        self.nodes = []
        self.variables = {}       # full name -> Variable
        self.var_order = []       # creation order
        self.scope = []           # stack of (name, reuse)


_G = _Graph()


def reset_default_graph():
    global _G
    _G = _Graph()


def set_random_seed(seed):
    return None


def _ex(x):
    return x._example if isinstance(x, Node) else x


class Node:
    def __init__(self, fn, inputs, name=None):
        self.fn = fn
        self.inputs = list(inputs)
        self.id = len(_G.nodes)
        self.name = name
        _G.nodes.append(self)
        self._example = self._compute([_ex(i) for i in self.inputs])

    def _compute(self, vals):
        return self.fn(*vals)

    @property
    def shape(self):
        return tuple(Dim(s) for s in self._example.shape)

    def get_shape(self):
        return self.shape

    # python operators used by the reference
    def __add__(self, o): return Node(lambda a, b: a + b, [self, o])
    def __radd__(self, o): return Node(lambda a, b: b + a, [self, o])
    def __sub__(self, o): return Node(lambda a, b: a - b, [self, o])
    def __rsub__(self, o): return Node(lambda a, b: b - a, [self, o])
    def __mul__(self, o): return Node(lambda a, b: a * b, [self, o])
    def __rmul__(self, o): return Node(lambda a, b: b * a, [self, o])
    def __truediv__(self, o): return Node(lambda a, b: a / b, [self, o])
    def __neg__(self): return Node(lambda a: -a, [self])

    def __getitem__(self, idx):
        return Node(lambda a: a[idx], [self])

    def __iter__(self):
        for i in range(self._example.shape[0]):
            yield self[i]

    def __len__(self):
        return self._example.shape[0]


class Placeholder(Node):
    def __init__(self, dtype, shape):
        self.dtype = dtype
        self._shape = [int(s) for s in shape]
        tdt = _FDT if dtype == float32 else torch.int64
        super().__init__(lambda: torch.zeros(self._shape, dtype=tdt), [])

    def convert(self, value):
        tdt = _FDT if self.dtype == float32 else torch.int64
        t = torch.as_tensor(np.asarray(value)).to(tdt)
        return t.reshape(self._shape)


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape if shape is not None else [])


class Variable(Node):
    def __init__(self, full_name, value):
        self.full_name = full_name
        self.value = torch.tensor(np.asarray(value), dtype=_FDT, requires_grad=True)
        super().__init__(lambda: self.value, [])

    @property
    def name(self):
        return self.full_name + ':0'

    @name.setter
    def name(self, v):
        pass

    def numpy(self):
        return self.value.detach().numpy().copy()

    def set(self, arr):
        with torch.no_grad():
            self.value.copy_(torch.as_tensor(np.asarray(arr), dtype=_FDT))

    def assign(self, other):
        op = _Op(lambda cache: self.set(_value_of(other, cache).detach().numpy()), [other])
        op.assign_target, op.assign_source = self, other
        return op


class _ScopeCtx:
    def __init__(self, name, reuse):
        self.name, self.reuse = name, reuse

    def __enter__(self):
        _G.scope.append((self.name, self.reuse))
        return self

    def __exit__(self, *a):
        _G.scope.pop()


def variable_scope(name, reuse=None):
    return _ScopeCtx(name, reuse)


def constant_initializer(value):
    def _init(shape, dtype=None, partition_info=None):
        return np.full(tuple(int(s) for s in shape), value, dtype=np.float32)
    return _init


def get_variable(name, shape, initializer=None, dtype=None):
    full = '/'.join([s for s, _ in _G.scope] + [name])
    reuse = any(r for _, r in _G.scope)
    if reuse:
        return _G.variables[full]
    if full in _G.variables:
        raise ValueError('Variable %s already exists' % full)
    shape = [int(s) for s in shape]
    v = Variable(full, initializer(shape, float32))
    _G.variables[full] = v
    _G.var_order.append(v)
    return v


def trainable_variables(scope=None):
    # TF: re.match(scope, name) -- a PREFIX match (SURVEY.md 8a footnote)
    import re
    return [v for v in _G.var_order if scope is None or re.match(scope, v.full_name)]


def global_variables():
    return list(_G.var_order)


def global_variables_initializer():
    return _Op(lambda cache: None, [])


# ---------------------------------------------------------------- ops
def matmul(a, b):
    return Node(lambda x, y: x @ y, [a, b])


def expand_dims(x, axis=None, dim=None):
    ax = axis if axis is not None else dim
    return Node(lambda t: t.unsqueeze(ax), [x])


def squeeze(x, axis=None):
    if axis is None:
        return Node(lambda t: t.squeeze(), [x])
    return Node(lambda t: t.squeeze(axis), [x])


def split(value=None, num_or_size_splits=None, axis=0, **kw):
    n = int(num_or_size_splits)
    size = int(_ex(value).shape[axis]) // n
    return [Node((lambda i: lambda t: torch.split(t, size, dim=axis)[i])(i), [value]) for i in range(n)]


def concat(values, axis=0):
    vals = list(values)
    return Node(lambda *ts: torch.cat([t if torch.is_tensor(t) else torch.as_tensor(t) for t in ts], dim=axis), vals)


def reshape(x, shape):
    shp = [int(s) for s in shape]
    return Node(lambda t: t.reshape(shp), [x])


def transpose(x, perm=None):
    if perm is None:
        return Node(lambda t: t.t() if t.dim() == 2 else t.permute(*reversed(range(t.dim()))), [x])
    return Node(lambda t: t.permute(*perm), [x])


def boolean_mask(x, mask):
    m = torch.as_tensor(np.asarray(mask).astype(np.bool_))
    return Node(lambda t: t[m], [x])


def reduce_mean(x, axis=None, keepdims=False):
    if axis is None:
        return Node(lambda t: t.mean(), [x])
    return Node(lambda t: t.mean(dim=axis, keepdim=keepdims), [x])


def reduce_sum(x, axis=None, keepdims=False):
    if axis is None:
        return Node(lambda t: t.sum(), [x])
    return Node(lambda t: t.sum(dim=axis, keepdim=keepdims), [x])


def square(x):
    return Node(lambda t: t * t, [x])


def log(x):
    return Node(torch.log, [x])


def tanh(x):
    return Node(torch.tanh, [x])


def clip_by_value(x, lo, hi):
    return Node(lambda t: torch.clamp(t, lo, hi), [x])


def argmax(x, axis=None):
    return Node(lambda t: torch.argmax(t) if axis is None else torch.argmax(t, dim=axis), [x])


def one_hot(x, depth, axis=-1):
    return Node(lambda t: torch.nn.functional.one_hot(t.long(), int(depth)).to(_FDT), [x])


def slice(x, begin, size):  # noqa: A001
    sl = tuple(np.s_[b:b + s] for b, s in zip(begin, size))
    return Node(lambda t: t[sl], [x])


def group(*ops):
    """tf.group of assigns: TF gives no ordering between the grouped reads and writes; this shim defines the
    update as SIMULTANEOUS -- every source is evaluated (and copied) before any target is written."""
    def run(cache):
        staged = []
        for o in ops:
            if hasattr(o, 'assign_target'):
                staged.append((o.assign_target, _value_of(o.assign_source, cache).detach().clone().numpy()))
            else:
                o.run(cache)
        for target, val in staged:
            target.set(val)
    return _Op(run, [])


class _nn:
    @staticmethod
    def relu(x):
        return Node(torch.relu, [x])

    @staticmethod
    def sigmoid(x):
        return Node(torch.sigmoid, [x])

    @staticmethod
    def tanh(x):
        return Node(torch.tanh, [x])

    @staticmethod
    def softmax(x):
        def f(t):
            e = torch.exp(t - t.max(dim=-1, keepdim=True).values)
            return e / e.sum(dim=-1, keepdim=True)
        return Node(f, [x])


nn = _nn()


# ---------------------------------------------------------------- evaluation
def _needed(fetch_nodes):
    seen, stack = set(), [n for n in fetch_nodes if isinstance(n, Node)]
    while stack:
        n = stack.pop()
        if n.id in seen:
            continue
        seen.add(n.id)
        for i in n.inputs:
            if isinstance(i, Node) and i.id not in seen:
                stack.append(i)
    return sorted(seen)


def _evaluate(fetch_nodes, cache):
    for nid in _needed(fetch_nodes):
        if nid in cache:
            continue
        n = _G.nodes[nid]
        if isinstance(n, Placeholder):
            raise ValueError('placeholder %d not fed' % nid)
        cache[nid] = n._compute([cache[i.id] if isinstance(i, Node) else i for i in n.inputs])


def _value_of(x, cache):
    if isinstance(x, Node):
        _evaluate([x], cache)
        return cache[x.id]
    return x


class _Op:
    """A stateful op (train step, assign, group)."""
    def __init__(self, fn, deps):
        self.fn, self.deps = fn, deps

    def run(self, cache):
        return self.fn(cache)


class _Grad(Node):
    """d loss / d var, produced by tf.gradients; all grads of one call share a holder."""
    def __init__(self, holder, k):
        self.holder, self.k = holder, k
        super().__init__(lambda: torch.zeros_like(holder['vars'][k].value), [])
        self.inputs = [holder['loss']]

    def _compute(self, vals):
        if not vals:      # build-time example
            return torch.zeros_like(self.holder['vars'][self.k].value)
        h = self.holder
        if h.get('loss_val') is not vals[0]:     # first grad node of this Session.run
            gs = torch.autograd.grad(vals[0], [v.value for v in h['vars']], retain_graph=True,
                                     allow_unused=True)
            h['grads'] = [torch.zeros_like(v.value) if g is None else g for g, v in zip(gs, h['vars'])]
            h['loss_val'] = vals[0]              # keeps the tensor alive: identity stays unique
        return h['grads'][self.k]


def gradients(loss, wts):
    holder = {'loss': loss, 'vars': list(wts)}
    return [_Grad(holder, k) for k in range(len(wts))]


def clip_by_global_norm(grads, clip_norm):
    def gnorm(*gs):
        return torch.sqrt(sum((g * g).sum() for g in gs))
    norm = Node(gnorm, grads)

    def scaled(g, n):
        scale = clip_norm * torch.minimum(1.0 / n, torch.tensor(1.0 / clip_norm, dtype=n.dtype))
        return g * scale
    return [Node(scaled, [g, norm]) for g in grads], norm


class _RMSProp:
    def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10):
        self.lr, self.decay, self.momentum, self.eps = learning_rate, decay, momentum, epsilon
        self.ms, self.mom = {}, {}

    def apply_gradients(self, grads_and_vars):
        gv = list(grads_and_vars)
        for _, v in gv:
            self.ms[v.full_name] = torch.ones_like(v.value.detach())
            self.mom[v.full_name] = torch.zeros_like(v.value.detach())

        def run(cache):
            lr = _value_of(self.lr, cache)
            gvals = [(_value_of(g, cache).detach(), v) for g, v in gv]   # all grads first, then apply
            with torch.no_grad():
                for g, v in gvals:
                    ms, mom = self.ms[v.full_name], self.mom[v.full_name]
                    ms += (g * g - ms) * (1 - self.decay)
                    mom.mul_(self.momentum).add_(lr * g / torch.sqrt(ms + self.eps))
                    v.value -= mom
        op = _Op(run, [g for g, _ in gv])
        op.is_train = True
        return op


class _Saver:
    def __init__(self, max_to_keep=5):
        pass

    def save(self, sess, path, global_step=None):
        torch.save({v.full_name: v.numpy() for v in _G.var_order}, '%s-%d.shim' % (path, global_step))

    def restore(self, sess, path):
        d = torch.load(path + '.shim', weights_only=False)
        for k, a in d.items():
            _G.variables[k].set(a)


class _train:
    RMSPropOptimizer = _RMSProp
    Saver = _Saver


train = _train()


class _Summary:
    @staticmethod
    def scalar(name, x):
        return Node(lambda t: t, [x], name=name)

    @staticmethod
    def merge(items):
        return Node(lambda *ts: torch.stack([torch.as_tensor(t, dtype=_FDT).reshape(()) for t in ts]), list(items))

    class FileWriter:
        def __init__(self, *a, **k):
            pass

        def add_summary(self, *a, **k):
            pass

        def flush(self):
            pass


summary = _Summary()


def ConfigProto(**kw):
    return None


class Session:
    def __init__(self, config=None):
        pass

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        cache = {}
        for ph, val in (feed_dict or {}).items():
            cache[ph.id] = ph.convert(val)
        # values first (pre-update), stateful ops afterwards -- like one TF step
        def value(f):              # nested fetch structures come back with the same nesting (tf.Session.run)
            if isinstance(f, (list, tuple)):
                return [value(g) for g in f]
            if isinstance(f, Node):
                o = _value_of(f, cache)
                return o.detach().numpy().copy() if torch.is_tensor(o) else o
            return None
        res = [value(f) for f in fl]
        for f in fl:
            if isinstance(f, _Op):
                f.run(cache)
        return res[0] if single else res
