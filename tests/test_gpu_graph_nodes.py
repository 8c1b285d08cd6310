"""What the captured hipGraphs hold, read back through the HIP runtime's graph API (tools/graph_nodes.py): every node of the
rollout graph, the update graph(s) and the test-episode graph of the three single-GPU BASELINE configs is a KERNEL node.

Why it is pinned: round 5's long-horizon determinism runs found a memset node inside a captured update that was not reliably
ordered in front of the kernel behind it on this stack (profiles/r05_determinism.txt; tools/memset_node_repro.py is the
stand-alone form).  aten's `copy_` of a contiguous tensor is a hipMemcpyAsync and `zero_()` may be a hipMemsetAsync: inside a
capture they become memcpy / memset nodes -- the trainer's capturable code issues neither (ops.copy_multi, `out=` forms,
pre-zeroed persistent buffers)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

from test_gpu_trainer import BASELINE_CASES, build  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('agent,scenario,E,n_step', BASELINE_CASES)
def test_captured_graphs_hold_kernel_nodes_only(agent, scenario, E, n_step):
    import graph_nodes as G
    env, model, tr = build(agent, E, True, scenario=scenario, n_step=n_step, keep_graphs=True)
    for _ in range(3):
        tr.run_batch()
    tr.evaluate(n_envs=32) if not env.name.startswith('atsc') else None
    torch.cuda.synchronize()
    assert tr.graph is not None and tr._upd is not None and tr.update_capture_error is None and tr.handoff_fallbacks == 0
    names = G.tensor_names(tr, model, model.policy, env, model.policy.params)
    graphs = {'rollout': tr.graph, 'update': tr._upd['grads']}
    if tr._upd['apply'] is not None:
        graphs['apply'] = tr._upd['apply']
    for ev in getattr(tr, '_eval_cache', {}).values():
        graphs['test episode'] = ev['graph']
    for what, g in graphs.items():
        c = G.census(g)
        assert c.get('kernel', 0) > 0
        assert set(c) == {'kernel'}, '%s graph of %s holds non-kernel nodes: %s\n%s' % (what, agent, c, G.describe(g, names))


def test_census_sees_memcpy_and_memset_nodes():
    """The census itself: a captured contiguous copy_ is a memcpy node, a captured hipMemsetAsync a memset node, a launch of
    this library a kernel node."""
    import graph_nodes as G
    from deeprl_network_amd import ops
    a, b = torch.zeros(1 << 16, device='cuda'), torch.ones(1 << 16, device='cuda')
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        a.copy_(b)
        G.hip_runtime().hipMemsetAsync(G.C.c_void_p(b.data_ptr()), 0, G.C.c_size_t(b.numel() * 4),
                                       G.C.c_void_p(torch.cuda.current_stream().cuda_stream))
        ops.copy_multi([(a, b)])
    c = G.census(g)
    assert c.get('kernel') == 1 and c.get('memcpy') == 1 and c.get('memset') == 1, c
    g.replay()
    torch.cuda.synchronize()
    assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0


def test_copy_multi_copies_every_pair():
    from deeprl_network_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(1)
    sizes = [1, 3, 16, 17, 4096, 100003, 1 << 20] + [5] * 12            # 19 pairs: two launches
    src = [torch.randint(0, 255, (n,), dtype=torch.uint8, device='cuda', generator=gen) for n in sizes]
    dst = [torch.zeros_like(s) for s in src]
    ops.copy_multi(zip(dst, src))
    for d, s in zip(dst, src):
        assert torch.equal(d, s)
    big = torch.randn(1000, 7, device='cuda', generator=gen)
    part_src, part_dst = big[3:900], torch.zeros(897, 7, device='cuda')  # (a contiguous view at an odd byte offset)
    ops.copy_multi([(part_dst, part_src), (dst[0], src[1][:1])])
    assert torch.equal(part_dst, part_src) and torch.equal(dst[0], src[1][:1])
    with pytest.raises(Exception):
        ops.copy_multi([(big[:, :3], big[:, 3:6])])
