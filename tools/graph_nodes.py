"""What a captured hipGraph holds: node census by type (kernel / memcpy / memset / ...) through the HIP runtime's own graph
API, with the destination of every memcpy / memset node named after the trainer tensor it falls into.

Library use (tests/test_gpu_graph_nodes.py):
    census(graph)                      -> {'kernel': n, 'memcpy': n, ...}    graph = torch.cuda.CUDAGraph(keep_graph=True)
    non_kernel_nodes(graph, names)     -> [(type, dst, bytes, tensor name or None), ...]

Command line: the node census of the rollout graph and of the update graph(s) of one BASELINE config, and -- with
--phases -- of each phase of the update captured on its own (which call issues the non-kernel node):
    python tools/graph_nodes.py [cacc|grid] [agent] [E] [--phases]
"""
import ctypes as C
import os
import sys

NODE_TYPES = ['kernel', 'memcpy', 'memset', 'host', 'graph', 'empty', 'wait_event', 'event_record', 'sem_signal', 'sem_wait',
              'mem_alloc', 'mem_free', 'memcpy_from_symbol', 'memcpy_to_symbol', 'batch_mem_op']


class _Pos(C.Structure):
    _fields_ = [('x', C.c_size_t), ('y', C.c_size_t), ('z', C.c_size_t)]


class _PitchedPtr(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('pitch', C.c_size_t), ('xsize', C.c_size_t), ('ysize', C.c_size_t)]


class _Memcpy3D(C.Structure):          # hipMemcpy3DParms (hip/driver_types.h)
    _fields_ = [('srcArray', C.c_void_p), ('srcPos', _Pos), ('srcPtr', _PitchedPtr), ('dstArray', C.c_void_p), ('dstPos', _Pos),
                ('dstPtr', _PitchedPtr), ('extent', _Pos), ('kind', C.c_int)]


class _Memset(C.Structure):            # hipMemsetParams (hip/hip_runtime_api.h)
    _fields_ = [('dst', C.c_void_p), ('elementSize', C.c_uint), ('height', C.c_size_t), ('pitch', C.c_size_t),
                ('value', C.c_uint), ('width', C.c_size_t)]


_hip = [None]


def hip_runtime():
    """The HIP runtime this process already mapped (torch's), by its path in /proc/self/maps: one runtime per process."""
    if _hip[0] is None:
        import torch  # noqa: F401
        path = None
        for line in open('/proc/self/maps'):
            if 'libamdhip64.so' in line:
                path = line.split()[-1]
                break
        if path is None:
            raise RuntimeError('libamdhip64.so is not mapped (no HIP torch?)')
        _hip[0] = C.CDLL(path)
    return _hip[0]


def _nodes(graph):
    hip = hip_runtime()
    g = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    rc = hip.hipGraphGetNodes(g, None, C.byref(n))
    if rc != 0:
        raise RuntimeError('hipGraphGetNodes -> %d' % rc)
    arr = (C.c_void_p * max(n.value, 1))()
    rc = hip.hipGraphGetNodes(g, arr, C.byref(n))
    if rc != 0:
        raise RuntimeError('hipGraphGetNodes -> %d' % rc)
    out = []
    for k in range(n.value):
        ty = C.c_int(-1)
        rc = hip.hipGraphNodeGetType(C.c_void_p(arr[k]), C.byref(ty))
        if rc != 0:
            raise RuntimeError('hipGraphNodeGetType -> %d' % rc)
        out.append((arr[k], ty.value))
    return out


def census(graph):
    """{node type name: count} of a torch.cuda.CUDAGraph captured with keep_graph=True."""
    c = {}
    for _, ty in _nodes(graph):
        name = NODE_TYPES[ty] if 0 <= ty < len(NODE_TYPES) else 'type_%d' % ty
        c[name] = c.get(name, 0) + 1
    return c


def tensor_names(*objs):
    """{name: tensor} of every CUDA tensor attribute (also inside lists / tuples / dicts, one level) of the given objects."""
    import torch
    names = {}
    for obj in objs:
        if obj is None:
            continue
        pre = type(obj).__name__
        for k, v in vars(obj).items():
            items = [(k, v)]
            if isinstance(v, (list, tuple)):
                items = [('%s[%d]' % (k, i), x) for i, x in enumerate(v)]
            elif isinstance(v, dict):
                items = [('%s[%r]' % (k, kk), x) for kk, x in v.items()]
            for kk, x in items:
                if torch.is_tensor(x) and x.is_cuda:
                    names['%s.%s' % (pre, kk)] = x
    return names


def _whose(addr, names):
    best = None
    for k, t in names.items():
        st = t.untyped_storage()
        lo = st.data_ptr()
        if lo <= addr < lo + st.nbytes() and (best is None or st.nbytes() < best[1]):
            best = (k, st.nbytes())
    return None if best is None else best[0]


def non_kernel_nodes(graph, names=None):
    """[(type name, dst address, bytes, name of the tensor the destination lies in)] of every node that is not a kernel."""
    hip = hip_runtime()
    names = names or {}
    out = []
    for node, ty in _nodes(graph):
        if ty == 0:
            continue
        name = NODE_TYPES[ty] if 0 <= ty < len(NODE_TYPES) else 'type_%d' % ty
        dst, nbytes = None, None
        if ty == 1:
            p = _Memcpy3D()
            if hip.hipGraphMemcpyNodeGetParams(C.c_void_p(node), C.byref(p)) == 0:
                dst = p.dstPtr.ptr
                nbytes = p.extent.x * max(p.extent.y, 1) * max(p.extent.z, 1)
        elif ty == 2:
            p = _Memset()
            if hip.hipGraphMemsetNodeGetParams(C.c_void_p(node), C.byref(p)) == 0:
                dst = p.dst
                nbytes = p.width * max(p.height, 1) * p.elementSize
        out.append((name, dst, nbytes, _whose(dst, names) if dst else None))
    return out


def describe(graph, names=None):
    c = census(graph)
    lines = ['  ' + ', '.join('%s %d' % kv for kv in sorted(c.items()))]
    for name, dst, nbytes, who in non_kernel_nodes(graph, names):
        lines.append('    %-8s -> %s  %s bytes  (%s)' % (name, hex(dst) if dst else '?', nbytes, who))
    return '\n'.join(lines)


def capture(fn):
    """fn captured on its own as a kept graph (no replay: capturing runs nothing)."""
    import torch
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        fn()
    return g


def trace_non_kernel(fn, roots=('deeprl_network_amd',)):
    """Capture fn() as a graph and attribute every non-kernel node to the Python source line (of a file whose path contains one
    of `roots`) that was executing when the node appeared: the capturing graph is polled from a line tracer through
    hipStreamGetCaptureInfo_v2.  -> (graph, [(file:line, node type name), ...])"""
    import torch
    hip = hip_runtime()
    found, state = [], {'seen': 0, 'last': None}

    def poll():
        st = C.c_int(0)
        gid = C.c_ulonglong(0)
        g = C.c_void_p(None)
        deps = C.c_void_p(None)
        nd = C.c_size_t(0)
        rc = hip.hipStreamGetCaptureInfo_v2(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(st), C.byref(gid),
                                            C.byref(g), C.byref(deps), C.byref(nd))
        if rc != 0 or st.value != 1 or not g.value:
            return None
        n = C.c_size_t(0)
        if hip.hipGraphGetNodes(g, None, C.byref(n)) != 0 or n.value == 0:
            return []
        arr = (C.c_void_p * n.value)()
        hip.hipGraphGetNodes(g, arr, C.byref(n))
        out = []
        for k in range(n.value):
            ty = C.c_int(-1)
            hip.hipGraphNodeGetType(C.c_void_p(arr[k]), C.byref(ty))
            if ty.value != 0:
                out.append(ty.value)
        return out

    def check(where):
        nk = poll()
        if nk is not None and len(nk) > state['seen']:
            for ty in nk[state['seen']:]:
                found.append((where, NODE_TYPES[ty] if 0 <= ty < len(NODE_TYPES) else 'type_%d' % ty))
            state['seen'] = len(nk)

    def local(frame, event, arg):
        if event in ('line', 'return'):
            if state['last'] is not None:
                check(state['last'])
            state['last'] = '%s:%d' % (os.path.relpath(frame.f_code.co_filename), frame.f_lineno)
        return local

    def tracer(frame, event, arg):
        if any(r in frame.f_code.co_filename for r in roots):
            return local(frame, event, arg)
        return None

    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    # (autograd's device thread is out of a line tracer's reach: the backward runs in the calling thread while traced)
    with torch.cuda.graph(g), torch.autograd.set_multithreading_enabled(False):
        sys.settrace(tracer)
        try:
            fn()
        finally:
            sys.settrace(None)
        if state['last'] is not None:
            check(state['last'])
    return g, found


def build_trainer(kind, agent, E):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    from helpers import cacc_config, grid_config
    from deeprl_network_amd.envs import make_batch_env
    from deeprl_network_amd.main import init_agent
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    if kind == 'grid':
        cp = grid_config(agent=agent)
    else:
        cp = cacc_config(agent=agent, scenario='slowdown' if agent == 'ma2c_nc' else 'catchup', n_step=60,
                         reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=E, device='cuda')
    np.random.seed(12)
    model = init_agent(env, cp['MODEL_CONFIG'], 10 ** 9, 12, num_envs=E, device='cuda')
    return env, model, BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True, keep_graphs=True)


def main():
    import torch
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    kind = args[0] if args else 'cacc'
    agent = args[1] if len(args) > 1 else ('ma2c_ic3' if kind == 'grid' else 'ia2c_fp')
    E = int(args[2]) if len(args) > 2 else (1024 if kind == 'grid' else 4096)
    env, model, tr = build_trainer(kind, agent, E)
    for _ in range(3):
        tr.run_batch()
    torch.cuda.synchronize()
    names = tensor_names(tr, model, model.policy, env, model.policy.params)
    print('%s %s E=%d' % (kind, agent, E))
    print(' rollout graph:\n' + describe(tr.graph, names))
    if tr._upd is None:
        print(' update: not captured (%s)' % tr.update_capture_error)
    else:
        print(' update graph (grads%s):\n' % ('' if tr._upd['apply'] is not None else ' + apply + epilogue') + describe(tr._upd['grads'], names))
        if tr._upd['apply'] is not None:
            print(' update graph (apply):\n' + describe(tr._upd['apply'], names))
    if '--phases' in sys.argv:
        m = model
        m.t = tr.n_step
        host = (m.policy._enc_was_saved, m.policy._mm_was_saved, m.policy._bits_steps)
        flags = getattr(tr, '_graph_flags', {})
        for k, v in flags.items():
            setattr(m.policy, k, v)
        phases = [('load_rewards', lambda: m.load_rewards(tr.buf_rraw)),
                  ('update_grads', lambda: m.update_grads(tr.R_end)),
                  ('update_apply', lambda: m.update_apply(0.0, rotate=False, lr_dev=tr.lr_dev)),
                  ('epilogue', tr._epilogue),
                  ('rollout', tr._rollout)]
        from deeprl_network_amd import ops
        # (trace_non_kernel runs autograd on THIS thread; its GEMM library handle must exist before a capture -- since round 6 the
        # update has no forward GEMM on the main thread that would have made one)
        with torch.autograd.set_multithreading_enabled(False):
            m.load_rewards(tr.buf_rraw)
            m.update_grads(tr.R_end)
        torch.cuda.synchronize()
        for k, v in flags.items():
            setattr(m.policy, k, v)
        keep = []
        ops.keepalive_begin(keep)
        try:
            for name, fn in phases:
                for k, v in flags.items():
                    setattr(m.policy, k, v)
                m.t = tr.n_step if name != 'rollout' else 0
                g, where = trace_non_kernel(fn)
                print(' phase %-13s\n%s' % (name, describe(g, names)))
                for w in where:
                    print('      %s node while executing %s' % (w[1], w[0]))
                keep.append(g)
        finally:
            ops.keepalive_end()
        m.policy._enc_was_saved, m.policy._mm_was_saved, m.policy._bits_steps = host


if __name__ == '__main__':
    main()
