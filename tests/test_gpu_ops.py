"""Per-kernel GPU parity (through the C-ABI) against the oracle restatements oracle/ops_ref.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _masks(kind):
    if kind == 'line':
        n = 8
        idx = np.arange(n)
        return (np.abs(idx[:, None] - idx[None, :]) == 1).astype(int)
    side = 5
    n = side * side
    d = np.array([[abs(i // side - j // side) + abs(i % side - j % side) for j in range(n)] for i in range(n)])
    return (d == 1).astype(int)


@pytest.mark.parametrize('kind', ['line', 'grid'])
@pytest.mark.parametrize('E,F', [(1, 5), (7, 4), (300, 64), (4096, 64), (33, 12)])
def test_nbr_gather_and_mean_fwd_bwd(kind, E, F):
    from deeprl_network_amd import ops
    from oracle import ops_ref
    nbr, cnt = ops.neighbor_table(_masks(kind), 'cuda')
    N = len(cnt)
    g = torch.Generator().manual_seed(E * 131 + F)
    x = torch.randn(N, E, F, generator=g)
    for fn, ref in [(ops.nbr_gather, ops_ref.nbr_gather), (ops.nbr_mean, ops_ref.nbr_mean)]:
        xg = x.cuda().requires_grad_(True)
        xc = x.clone().requires_grad_(True)
        y = fn(xg, nbr)
        yr = ref(xc, nbr.cpu())
        assert torch.equal(y.cpu(), yr) if fn is ops.nbr_gather else torch.allclose(y.cpu(), yr, rtol=1e-6, atol=1e-7)
        w = torch.randn(yr.shape, generator=g)
        (y * w.cuda()).sum().backward()
        (yr * w).sum().backward()
        torch.testing.assert_close(xg.grad.cpu(), xc.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('kind,A', [('line', 4), ('grid', 5)])
def test_nbr_onehot(kind, A):
    from deeprl_network_amd import ops
    from oracle import ops_ref
    nbr, cnt = ops.neighbor_table(_masks(kind), 'cuda')
    N = len(cnt)
    for E in (1, 50, 4096):
        a = torch.randint(0, A, (E, N), dtype=torch.uint8)
        y = ops.nbr_onehot(a.cuda(), nbr, A)
        assert torch.equal(y.cpu(), ops_ref.nbr_onehot(a, nbr.cpu(), A))


@pytest.mark.parametrize('N,E,H', [(8, 1, 64), (8, 257, 64), (25, 64, 64), (8, 4096, 64), (3, 5, 16)])
def test_lstm_cell_fwd_bwd(N, E, H):
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N * E + H)
    z = torch.randn(N, E, 4 * H, generator=g) * 2
    b = torch.randn(N, 4 * H, generator=g) * 0.3
    c = torch.randn(N, E, H, generator=g)
    done = (torch.rand(E, generator=g) < 0.3).float()
    ins_g = [t.cuda().requires_grad_(True) for t in (z, b, c)]
    ins_c = [t.double().requires_grad_(True) for t in (z, b, c)]
    h1, c1 = ops.lstm_cell(ins_g[0], ins_g[1], ins_g[2], done.cuda())
    h2, c2 = ops_ref.lstm_cell(ins_c[0], ins_c[1], ins_c[2], done.double())
    torch.testing.assert_close(h1.cpu().double(), h2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(c1.cpu().double(), c2, rtol=1e-5, atol=1e-6)
    wh, wc = torch.randn(N, E, H, generator=g), torch.randn(N, E, H, generator=g)
    ((h1 * wh.cuda()).sum() + (c1 * wc.cuda()).sum()).backward()
    ((h2 * wh.double()).sum() + (c2 * wc.double()).sum()).backward()
    for a, r, name in zip(ins_g, ins_c, 'zbc'):
        torch.testing.assert_close(a.grad.cpu().double(), r.grad, rtol=2e-4, atol=2e-5 * max(1, E ** 0.5), msg=name)
    # strided bias view (row stride > 4H), as handed over by the flat parameter buffer
    big = torch.zeros(N, 4 * H + 40, device='cuda')        # 16-byte aligned rows, like the ParamStore views
    big[:, 8:8 + 4 * H] = b.cuda()
    h3, c3 = ops.lstm_cell(z.cuda(), big[:, 8:8 + 4 * H], c.cuda(), done.cuda())
    assert torch.equal(h3, h1.detach()) and torch.equal(c3, c1.detach())


@pytest.mark.parametrize('N,T,E,H', [(8, 6, 33, 64), (3, 4, 128, 16)])
def test_lstm_sequence_fwd_bwd(N, T, E, H):
    """Fused recurrence (one wgrad GEMM / one bias reduction) == plain per-step autograd loop."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(7)
    pre = torch.randn(N, T, E, 4 * H, generator=g)
    wh = torch.randn(N, H, 4 * H, generator=g) * 0.2
    b = torch.randn(N, 4 * H, generator=g) * 0.1
    h0, c0 = torch.randn(N, E, H, generator=g) * 0.5, torch.randn(N, E, H, generator=g) * 0.5
    done = (torch.rand(T, E, generator=g) < 0.2).float()
    w = torch.randn(N, T, E, H, generator=g)
    ins_g = [t.cuda().requires_grad_(True) for t in (pre, wh, b, h0, c0)]
    ins_c = [t.double().requires_grad_(True) for t in (pre, wh, b, h0, c0)]
    Hg = ops.lstm_sequence(*ins_g, done.cuda())
    Hc = ops_ref.lstm_sequence(*ins_c, done.double())
    torch.testing.assert_close(Hg.cpu().double(), Hc, rtol=1e-4, atol=1e-5)
    (Hg * w.cuda()).sum().backward()
    (Hc * w.double()).sum().backward()
    for a, r, name in zip(ins_g, ins_c, ['pre', 'wh', 'b', 'h0', 'c0']):
        torch.testing.assert_close(a.grad.cpu().double(), r.grad, rtol=1e-3, atol=2e-4, msg=name)


def test_bias_act_and_cell_second_addend():
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(11)
    N, E, H = 8, 77, 64
    for act in (ops.BIAS_NONE, ops.BIAS_RELU, ops.BIAS_TANH):
        x = torch.randn(N, E, H, generator=g)
        b = torch.randn(N, H, generator=g)
        y = ops.bias_act_(x.clone().cuda(), b.cuda(), act)
        torch.testing.assert_close(y.cpu(), ops_ref.bias_act_(x.clone(), b, act), rtol=1e-6, atol=1e-6)
    # out = column block of a wider buffer (concat without copy)
    x = torch.randn(N, E, H, generator=g)
    b = torch.randn(N, H, generator=g)
    wide = torch.zeros(N, E, 2 * H, device='cuda')
    ops.bias_act_(x.cuda(), b.cuda(), ops.BIAS_RELU, out=wide[:, :, H:])
    assert torch.all(wide[:, :, :H] == 0)
    torch.testing.assert_close(wide[:, :, H:].cpu(), torch.relu(x + b[:, None]), rtol=1e-6, atol=1e-6)
    z, z2 = torch.randn(N, E, 4 * H, generator=g), torch.randn(N, E, 4 * H, generator=g)
    b = torch.randn(N, 4 * H, generator=g)
    c = torch.randn(N, E, H, generator=g)
    done = (torch.rand(E, generator=g) < 0.3).float()
    co, ho = torch.empty(N, E, H, device='cuda'), torch.empty(N, E, H, device='cuda')
    ops.lstm_cell_infer(z.cuda(), b.cuda(), c.cuda(), done.cuda(), co, ho, z2=z2.cuda())
    hr, cr = ops_ref.lstm_cell(z + z2, b, c, done)
    torch.testing.assert_close(ho.cpu(), hr, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(co.cpu(), cr, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('N,E', [(8, 4096), (8, 1), (25, 130), (3, 127), (8, 257)])
@pytest.mark.parametrize('two_addends', [False, True])
def test_lstm_step_fused_mfma(N, E, two_addends):
    """MFMA GEMM + cell in one kernel == (h*(1-done)) @ Wh + addends -> cell, incl. ragged row counts, strided
    sequence slots, gates output and in-place state update."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    g = torch.Generator().manual_seed(N * 1000 + E)
    h = torch.randn(N, E, H, generator=g) * 0.7
    c = torch.randn(N, E, H, generator=g)
    z1 = torch.randn(N, E, 4 * H, generator=g)
    z2 = torch.randn(N, E, 4 * H, generator=g) if two_addends else None
    done = (torch.rand(E, generator=g) < 0.3).float()
    # asymmetric weights (a swapped A/B or row/col mapping cannot pass) inside a wider flat row, like ParamStore
    flat = torch.zeros(N, H * 4 * H + 4 * H + 48)
    wh = flat[:, 16:16 + H * 4 * H].view(N, H, 4 * H)
    wh.copy_(torch.randn(N, H, 4 * H, generator=g) * 0.2 + torch.arange(4 * H).view(1, 1, -1) * 1e-3)
    b = flat[:, 16 + H * 4 * H + 16:16 + H * 4 * H + 16 + 4 * H]
    b.copy_(torch.randn(N, 4 * H, generator=g) * 0.1)
    hr, cr = torch.empty(N, E, H, dtype=torch.float64), torch.empty(N, E, H, dtype=torch.float64)
    ops_ref.lstm_step_fused(h.double(), wh.double(), b.double(), z1.double(), None if z2 is None else z2.double(),
                            c.double(), done.double(), None, cr, hr)
    fg = flat.cuda()
    whg = fg[:, 16:16 + H * 4 * H].view(N, H, 4 * H)
    bg = fg[:, 16 + H * 4 * H + 16:16 + H * 4 * H + 16 + 4 * H]
    # outputs into slot 1 of [N,3,E,*] sequence buffers, gates requested
    Hbuf = torch.zeros(N, 3, E, H, device='cuda'); Cbuf = torch.zeros(N, 3, E, H, device='cuda')
    Gbuf = torch.zeros(N, 3, E, 4 * H, device='cuda')
    Hbuf[:, 0].copy_(h); Cbuf[:, 0].copy_(c)
    ops.lstm_step_fused(Hbuf[:, 0], whg, bg, z1.cuda(), None if z2 is None else z2.cuda(), Cbuf[:, 0], done.cuda(),
                        Gbuf[:, 1], Cbuf[:, 1], Hbuf[:, 1])
    torch.testing.assert_close(Hbuf[:, 1].cpu().double(), hr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(Cbuf[:, 1].cpu().double(), cr, rtol=2e-5, atol=2e-6)
    assert torch.all(Hbuf[:, 2] == 0) and torch.all(Gbuf[:, 2] == 0) and torch.all(Gbuf[:, 0] == 0)
    gi = Gbuf[:, 1, :, :H].cpu().double()
    assert torch.all((gi > 0) & (gi < 1))
    # in place (rollout): h_out aliases h, c_out aliases c, no gates
    hg, cg = h.cuda(), c.cuda()
    ops.lstm_step_fused(hg, whg, bg, z1.cuda(), None if z2 is None else z2.cuda(), cg, done.cuda(), None, cg, hg)
    torch.testing.assert_close(hg.cpu().double(), hr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(cg.cpu().double(), cr, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('N,E,A,m_max', [(8, 4096, 4, 2), (25, 130, 5, 4), (3, 127, 8, 2), (8, 1, 4, 2)])
@pytest.mark.parametrize('two_addends', [False, True])
@pytest.mark.parametrize('mode', [0, 1, 2])
def test_lstm_step_fused_heads(N, E, A, m_max, two_addends, mode):
    """The actor / critic epilogues of the fused step (forward 'p' + draw, forward 'v') == step, then
    softmax(h' W + b) + sample_actions, resp. [h', onehot(nbr actions)] W + b, incl. ragged rows, in-place
    state, parameters living in a wider flat row, -1 padded neighbour tables and all draw modes."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    g = torch.Generator().manual_seed(N * 1000 + E + 7 * A)
    h = torch.randn(N, E, H, generator=g) * 0.7
    c = torch.randn(N, E, H, generator=g)
    z1 = torch.randn(N, E, 4 * H, generator=g)
    z2 = torch.randn(N, E, 4 * H, generator=g) if two_addends else None
    done = (torch.rand(E, generator=g) < 0.3).float()
    n_na = m_max * A
    offs, sizes, o = {}, dict(wh=H * 4 * H, b=4 * H, pi_w=H * A, pi_b=A, v_w=H + n_na, v_b=1), 16
    for k, n in sizes.items():
        offs[k] = o
        o += (n + 15) // 16 * 16
    flat = torch.zeros(N, o + 16)
    view = lambda f, k, *shape: f[:, offs[k]:offs[k] + sizes[k]].view(N, *shape)      # noqa: E731
    view(flat, 'wh', H, 4 * H).copy_(torch.randn(N, H, 4 * H, generator=g) * 0.2)
    view(flat, 'b', 4 * H).copy_(torch.randn(N, 4 * H, generator=g) * 0.1)
    view(flat, 'pi_w', H, A).copy_(torch.randn(N, H, A, generator=g) * 0.5)
    view(flat, 'pi_b', A).copy_(torch.randn(N, A, generator=g) * 0.3)
    view(flat, 'v_w', H + n_na, 1).copy_(torch.randn(N, H + n_na, 1, generator=g))
    view(flat, 'v_b', 1).copy_(torch.randn(N, 1, generator=g))
    # ragged neighbour table: agent i has (i % m_max) + 1 neighbours, ascending, -1 padded
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        others = [j for j in range(N) if j != i][:(i % m_max) + 1]
        idx[i, :len(others)] = torch.tensor(others, dtype=torch.int32)
    u = torch.rand(E, N, generator=g)
    draw = dict(mode=mode, u=u if mode == 0 else None, seed=77, env_id_base=1000, step=5)
    step_dev = torch.tensor(37, dtype=torch.int64)
    d64 = lambda t: None if t is None else t.double()                                  # noqa: E731
    f64 = flat.double()
    hr, cr = torch.empty(N, E, H, dtype=torch.float64), torch.empty(N, E, H, dtype=torch.float64)
    pir, actr = torch.empty(N, E, A, dtype=torch.float64), torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.lstm_step_policy(h.double(), view(f64, 'wh', H, 4 * H), view(f64, 'b', 4 * H), z1.double(), d64(z2), c.double(),
                             done.double(), cr, hr, view(f64, 'pi_w', H, A), view(f64, 'pi_b', A), pir, actr,
                             step_dev=step_dev, **draw)
    fg = flat.cuda()
    hg, cg = h.cuda(), c.cuda()
    pig, actg = torch.zeros(N, 2, E, A, device='cuda'), torch.full((E, N), 255, dtype=torch.uint8, device='cuda')
    dg = dict(draw, u=None if draw['u'] is None else u.cuda())
    z2g = None if z2 is None else z2.cuda()
    ops.lstm_step_policy(hg, view(fg, 'wh', H, 4 * H), view(fg, 'b', 4 * H), z1.cuda(), z2g, cg, done.cuda(), cg, hg,
                         view(fg, 'pi_w', H, A), view(fg, 'pi_b', A), pig[:, 1], actg, step_dev=step_dev.cuda(), **dg)
    torch.testing.assert_close(hg.cpu().double(), hr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(cg.cpu().double(), cr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(pig[:, 1].cpu().double(), pir, rtol=2e-5, atol=2e-6)
    assert torch.all(pig[:, 0] == 0)
    torch.testing.assert_close(pig[:, 1].sum(-1).cpu(), torch.ones(N, E), rtol=0, atol=1e-6)
    # the draw, bit-exact given the kernel's own probabilities (a cdf boundary within 1 ulp of u may differ from the
    # float64 softmax' draw, which is the case for ~1e-7 of the samples)
    act_chk = torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.sample_actions(pig[:, 1].cpu(), act_chk, step_dev=step_dev, **draw)
    assert torch.equal(actg.cpu(), act_chk)
    assert (actg.cpu() != actr).float().mean() < 1e-3
    # critic re-step from the new state (quirk Q1), out of place
    vr = torch.empty(N, E, dtype=torch.float64)
    h2r, c2r = torch.empty_like(hr), torch.empty_like(cr)
    ops_ref.lstm_step_value(hr, view(f64, 'wh', H, 4 * H), view(f64, 'b', 4 * H), z1.double(), d64(z2), cr, done.double(),
                            c2r, h2r, view(f64, 'v_w', H + n_na, 1), view(f64, 'v_b', 1), act_chk, idx, A, vr)
    vbuf = torch.zeros(3, N, E, device='cuda')
    h2, c2 = torch.zeros_like(hg), torch.zeros_like(cg)
    ops.lstm_step_value(hg, view(fg, 'wh', H, 4 * H), view(fg, 'b', 4 * H), z1.cuda(), z2g, cg, done.cuda(), c2, h2,
                        view(fg, 'v_w', H + n_na, 1), view(fg, 'v_b', 1), actg, idx.cuda(), A, vbuf[1])
    torch.testing.assert_close(h2.cpu().double(), h2r, rtol=5e-5, atol=5e-6)
    torch.testing.assert_close(vbuf[1].cpu().double(), vr, rtol=1e-4, atol=2e-5)
    assert torch.all(vbuf[0] == 0) and torch.all(vbuf[2] == 0)


@pytest.mark.parametrize('N,E,A,m_max', [(8, 4096, 4, 2), (25, 130, 5, 4), (3, 127, 8, 2), (8, 1, 4, 2)])
@pytest.mark.parametrize('two_addends', [False, True])
@pytest.mark.parametrize('mode', [1, 2])
def test_lstm_step_policy_value_one_kernel(N, E, A, m_max, two_addends, mode):
    """forward('p') + forward('v') of a lock-step in one kernel (+ the neighbour-action add) == policy step, then the
    value re-step from the new state (float64), with some replicas starting an episode (done = 1 masks BOTH steps)."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    g = torch.Generator().manual_seed(N * 77 + E + A)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    h, c, z1 = r(N, E, H) * 0.7, r(N, E, H), r(N, E, 4 * H)
    z2 = r(N, E, 4 * H) if two_addends else None
    done = (torch.rand(E, generator=g) < 0.3).float()
    wh, b = r(N, H, 4 * H) * 0.2, r(N, 4 * H) * 0.1
    pi_w, pi_b = r(N, H, A) * 0.5, r(N, A) * 0.3
    v_w, v_b = r(N, H + m_max * A, 1), r(N, 1)
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        others = [j for j in range(N) if j != i][:(i % m_max) + 1]
        idx[i, :len(others)] = torch.tensor(others, dtype=torch.int32)
    d = lambda t: None if t is None else t.double()                                     # noqa: E731
    cu = lambda t: None if t is None else t.cuda()                                      # noqa: E731
    draw = dict(mode=mode, seed=5, env_id_base=40, step=3)
    # product
    hg, cg = h.cuda(), c.cuda()
    pig, actg = torch.zeros(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda')
    vg = torch.zeros(N, E, device='cuda')
    ops.lstm_step_policy_value(hg, cu(wh), cu(b), cu(z1), cu(z2), cg, cu(done), cu(pi_w), cu(pi_b), pig, actg, cu(v_w), cu(v_b),
                               cu(idx), A, vg, **draw)
    # oracle: the policy half in float64, the draw from the kernel's own probabilities, then the value half
    hr, cr = h.double(), c.double()
    pir, actr = torch.zeros(N, E, A, dtype=torch.float64), torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.lstm_step_policy(hr, d(wh), d(b), d(z1), d(z2), cr, d(done), cr, hr, d(pi_w), d(pi_b), pir, actr, **draw)
    torch.testing.assert_close(hg.cpu().double(), hr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(cg.cpu().double(), cr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(pig.cpu().double(), pir, rtol=2e-5, atol=2e-6)
    act_chk = torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.sample_actions(pig.cpu(), act_chk, **draw)
    assert torch.equal(actg.cpu(), act_chk)
    vr = torch.zeros(N, E, dtype=torch.float64)
    ops_ref.lstm_step_value(hr, d(wh), d(b), d(z1), d(z2), cr, d(done), torch.empty_like(cr), torch.empty_like(hr), d(v_w),
                            d(v_b), act_chk, idx, A, vr)
    torch.testing.assert_close(vg.cpu().double(), vr, rtol=1e-4, atol=3e-5)


def test_lstm_step_fused_head_rejects_wide_action_sets():
    from deeprl_network_amd import _lib, ops
    N, E, H, A = 2, 4, 64, 9
    z = lambda *s: torch.zeros(*s, device='cuda')                                       # noqa: E731
    with pytest.raises(_lib.NmarlError):
        ops.lstm_step_policy(z(N, E, H), z(N, H, 4 * H), z(N, 4 * H), z(N, E, 4 * H), None, z(N, E, H), z(E), z(N, E, H),
                             z(N, E, H), z(N, H, A), z(N, A), z(N, E, A), torch.zeros(E, N, dtype=torch.uint8, device='cuda'),
                             mode=2)


@pytest.mark.parametrize('N,rows,F', [(8, 4096, 15), (8, 4103, 8), (25, 130, 60), (25, 1, 20), (3, 257, 64), (8, 70000, 15)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_fc_small_layer_fwd_bwd(N, rows, F, act):
    """csrc/fc.hip: act(x w + b) and its (dw, db) for small input widths == the float64 restatement, with x read in
    place from the env-major slab (transposed view), y / dy as column blocks of a wider buffer, parameters as
    views of a flat row, ragged row counts and multi-chunk partial sums."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N * 31 + rows + F)
    slab = torch.randn(rows, N, F, generator=g)
    x = slab.transpose(0, 1)                                   # [N,rows,F], strides (F, N*F, 1)
    flat = torch.zeros(N, 16 + F * 64 + 16 + 64 + 16)
    w = flat[:, 16:16 + F * 64].view(N, F, 64)
    b = flat[:, 32 + F * 64:32 + F * 64 + 64]
    w.copy_(torch.randn(N, F, 64, generator=g) * 0.4)
    b.copy_(torch.randn(N, 64, generator=g) * 0.2)
    yr = ops_ref.fc_fwd(x.double(), w.double(), b.double(), act)
    fg, sg = flat.cuda(), slab.cuda()
    xg = sg.transpose(0, 1)
    wg, bg = fg[:, 16:16 + F * 64].view(N, F, 64), fg[:, 32 + F * 64:32 + F * 64 + 64]
    S = torch.zeros(N, rows, 192, device='cuda')
    ops.fc_fwd(xg, wg, bg, act, out=S[:, :, 64:128])
    torch.testing.assert_close(S[:, :, 64:128].cpu().double(), yr, rtol=1e-5, atol=1e-5)
    assert torch.all(S[:, :, :64] == 0) and torch.all(S[:, :, 128:] == 0)
    torch.testing.assert_close(ops.fc_fwd(xg, wg, bg, act).cpu().double(), yr, rtol=1e-5, atol=1e-5)
    dS = torch.randn(N, rows, 192, generator=g)
    # the activation derivative is a function of the layer OUTPUT: take the kernel's own y (a pre-activation within
    # 1e-7 of zero may round to either side of the relu kink in fp32 vs float64)
    dwr, dbr = ops_ref.fc_bwd(x.double(), S[:, :, 64:128].cpu().double(), dS[:, :, 64:128].double(), act)
    dw, db = ops.fc_bwd(xg, S[:, :, 64:128], dS.cuda()[:, :, 64:128], act)
    scale = float(rows) ** 0.5
    torch.testing.assert_close(dw.cpu().double(), dwr, rtol=1e-4, atol=2e-5 * scale)
    torch.testing.assert_close(db.cpu().double(), dbr, rtol=1e-4, atol=2e-5 * scale)
    # determinism: fixed-order partial sums
    dw2, db2 = ops.fc_bwd(xg, S[:, :, 64:128], dS.cuda()[:, :, 64:128], act)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize('N,rows,Fx,A,m_max', [(8, 4096, 15, 4, 2), (25, 1030, 60, 5, 4), (3, 1, 5, 4, 2)])
def test_fc_fwd_multi_with_gathered_fingerprints(N, rows, Fx, A, m_max):
    """Both encoder layers of a lock-step in one launch, the second reading the previous-step policies through the
    neighbour table == gather + two fc layers + concat (float64)."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N + rows + Fx)
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        others = [j for j in range(N) if j != i][:(i % m_max) + 1]
        idx[i, :len(others)] = torch.tensor(others, dtype=torch.int32)
    slab = torch.randn(rows, N, Fx, generator=g)
    fp = torch.softmax(torch.randn(N, rows, A, generator=g), -1)
    w1, b1 = torch.randn(N, Fx, 64, generator=g) * 0.3, torch.randn(N, 64, generator=g) * 0.1
    w2, b2 = torch.randn(N, m_max * A, 64, generator=g) * 0.3, torch.randn(N, 64, generator=g) * 0.1
    d = lambda t: t.double()                                                             # noqa: E731
    ref = ops_ref.fc_fwd_multi([(d(slab.transpose(0, 1)), d(w1), d(b1), None), (d(fp), d(w2), d(b2), idx)], 1)
    c = lambda t: t.cuda()                                                               # noqa: E731
    got = ops.fc_fwd_multi([(c(slab).transpose(0, 1), c(w1), c(b1), None), (c(fp), c(w2), c(b2), c(idx))], 1)
    assert got.shape == (N, rows, 128)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-5, atol=1e-5)


def test_fc_concat_autograd_matches_torch():
    from deeprl_network_amd import ops
    N, rows = 8, 1000
    g = torch.Generator().manual_seed(5)
    x1, x2 = torch.randn(N, rows, 15, generator=g).cuda(), torch.randn(N, rows, 8, generator=g).cuda()
    ps = [(torch.randn(N, F, 64, generator=g) * 0.3).cuda().requires_grad_() for F in (15, 8)]
    bs = [(torch.randn(N, 64, generator=g) * 0.1).cuda().requires_grad_() for _ in range(2)]
    wx = (torch.randn(N, 128, 32, generator=g) * 0.1).cuda()
    R = torch.randn(N, rows, 32, generator=g).cuda()
    s = ops.fc_concat([(x1, ps[0], bs[0]), (x2, ps[1], bs[1])], ops.BIAS_RELU)
    (torch.bmm(s, wx) * R).sum().backward()
    got = [t.grad.clone() for t in ps + bs]
    for t in ps + bs:
        t.grad = None
    ref = torch.cat([torch.relu(torch.baddbmm(bs[i].unsqueeze(1), x, ps[i])) for i, x in enumerate((x1, x2))], dim=-1)
    torch.testing.assert_close(s, ref, rtol=1e-5, atol=1e-5)
    (torch.bmm(ref, wx) * R).sum().backward()
    for a, t in zip(got, ps + bs):
        torch.testing.assert_close(a, t.grad, rtol=1e-4, atol=1e-4)
    # inputs wider than 64: the plain GEMM path, same result
    xw = torch.randn(N, rows, 128, generator=g).cuda()
    ww, bw = (torch.randn(N, 128, 64, generator=g) * 0.1).cuda(), torch.zeros(N, 64).cuda()
    torch.testing.assert_close(ops.fc_concat([(xw, ww, bw)], ops.BIAS_TANH), torch.tanh(torch.bmm(xw, ww)), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('N,rows,A,m_max,act', [(8, 4096 * 3 + 17, 5, 3, 1), (25, 1300, 12, 5, 2), (5, 63, 4, 2, 1)])
def test_fc_bwd_with_in_kernel_gather(N, rows, A, m_max, act):
    """nmarl_fc_bwd_gather: the layer's input gathered over a (-1 padded) neighbour table inside the backward kernel, read
    in place from an env-major slab [rows,N,A] -- vs the same kernel on the materialised gather (bit for bit) and vs the
    float64 restatement; S / dS as column blocks of wider buffers (the update's layout)."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N * 3 + rows + A)
    slab = torch.randn(rows, N, A, generator=g)                       # env-major: agent stride A, row pitch N*A
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        js = [j for j in (i, i - 1, i + 1, i + 2, i - 2) if 0 <= j < N][:(i % m_max) + 1]
        idx[i, :len(js)] = torch.tensor(js, dtype=torch.int32)
    F = A * m_max
    S = torch.relu(torch.randn(N, rows, 128, generator=g)) if act == 1 else torch.tanh(torch.randn(N, rows, 128, generator=g))
    dS = torch.randn(N, rows, 128, generator=g)
    xv = slab.cuda().transpose(0, 1)                                  # [N,rows,A] view of the slab
    Sg, dSg, idxg = S.cuda(), dS.cuda(), idx.cuda()
    dw, db = ops.fc_bwd(xv, Sg[:, :, 64:], dSg[:, :, 64:], act, nbr_idx=idxg)
    xg = ops.nbr_gather(xv.contiguous(), idxg)
    assert xg.shape == (N, rows, F)
    dw2, db2 = ops.fc_bwd(xg, Sg[:, :, 64:], dSg[:, :, 64:], act)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dwr, dbr = ops_ref.fc_bwd(slab.double().transpose(0, 1), S[:, :, 64:].double(), dS[:, :, 64:].double(), act, nbr_idx=idx)
    scale = max(1.0, float(dwr.abs().max()))
    torch.testing.assert_close(dw.cpu().double(), dwr, rtol=2e-4, atol=2e-5 * scale)
    torch.testing.assert_close(db.cpu().double(), dbr, rtol=2e-4, atol=2e-5 * scale)


@pytest.mark.parametrize('N,rows,O', [(8, 4096, 5), (25, 131, 6), (3, 1, 1), (8, 70001, 5), (2, 300, 8)])
def test_thin_linear_bwd(N, rows, O):
    """Heads backward in one streaming pass == dy w^T, h^T dy, sum dy (float64), deterministic."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N + rows + O)
    h, dy = torch.randn(N, rows, 64, generator=g), torch.randn(N, rows, O, generator=g)
    w = torch.randn(N, 64, O, generator=g)
    dhr, dwr, dbr = ops_ref.thin_linear_bwd(h.double(), dy.double(), w.double())
    dh, dw, db = ops.thin_linear_bwd(h.cuda(), dy.cuda(), w.cuda())
    scale = float(rows) ** 0.5
    torch.testing.assert_close(dh.cpu().double(), dhr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dw.cpu().double(), dwr, rtol=1e-4, atol=2e-5 * scale)
    torch.testing.assert_close(db.cpu().double(), dbr, rtol=1e-4, atol=2e-5 * scale)
    dh2, dw2, db2 = ops.thin_linear_bwd(h.cuda(), dy.cuda(), w.cuda())
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dh, dh2)
    # h as the slots 1.. of a [N,T+1,E,64] sequence buffer (agent-strided panels), read in place
    if rows % 2 == 0:
        buf = torch.zeros(N, rows // 2 * 3, 64, device='cuda')
        hv = buf[:, rows // 2:]
        hv.copy_(h)
        assert not hv.is_contiguous()
        dh4, dw4, db4 = ops.thin_linear_bwd(hv, dy.cuda(), w.cuda())
        assert torch.equal(dh4, dh) and torch.equal(dw4, dw) and torch.equal(db4, db)
    if O > 1:    # last column's gradient handed over separately (actor logits + critic value)
        dh3, dw3, db3 = ops.thin_linear_bwd(h.cuda(), dy[..., :O - 1].cuda(), w.cuda(), dy2=dy[..., O - 1].cuda())
        assert torch.equal(dh3, dh) and torch.equal(dw3, dw) and torch.equal(db3, db)
    # through autograd
    hg, wg, bg = h.cuda().requires_grad_(), w.cuda().requires_grad_(), torch.zeros(N, O).cuda().requires_grad_()
    (ops.thin_linear(hg, wg, bg) * dy.cuda()).sum().backward()
    torch.testing.assert_close(hg.grad, dh)
    torch.testing.assert_close(wg.grad, dw)
    torch.testing.assert_close(bg.grad, db)


@pytest.mark.parametrize('N,rows,A,m_max', [(8, 4096, 4, 2), (25, 1000, 5, 4), (3, 70001, 4, 2), (5, 1, 7, 4)])
def test_heads_and_neighbour_action_value(N, rows, A, m_max):
    """ops.heads (skinny GEMM + gathered neighbour-action term; streaming backward + histogram) == the plain
    autograd restatement with an explicit one-hot, values and all five gradients."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N * 7 + rows + A)
    H = 64
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        others = [j for j in range(N) if j != i][:(i % m_max) + 1]
        idx[i, :len(others)] = torch.tensor(others, dtype=torch.int32)
    act = torch.randint(0, A, (rows, N), generator=g).to(torch.uint8)
    h = torch.randn(N, rows, H, generator=g)
    prm = [torch.randn(N, H, A, generator=g) * 0.3, torch.randn(N, A, generator=g) * 0.1,
           torch.randn(N, H + m_max * A, 1, generator=g) * 0.3, torch.randn(N, 1, generator=g)]
    R1, R2 = torch.randn(N, rows, A, generator=g), torch.randn(N, rows, generator=g)
    w_a = prm[2][:, H:]
    torch.testing.assert_close(ops.nbr_action_value(act.cuda(), idx.cuda(), w_a.cuda(), A).cpu().double(),
                               ops_ref.nbr_action_value(act, idx, w_a.double(), A), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ops.nbr_action_value_bwd(act.cuda(), idx.cuda(), R2.cuda(), A).cpu().double(),
                               ops_ref.nbr_action_value_bwd(act, idx, R2.double(), A), rtol=1e-4, atol=2e-5 * rows ** 0.5)

    def run(mod, tensors, dev):
        ts = [t.to(dev).clone().requires_grad_() for t in tensors]
        logits, v = mod.heads(ts[0], ts[1], ts[2], ts[3], ts[4], act.to(dev), idx.to(dev), A)
        ((torch.softmax(logits, -1) * R1.to(ts[0])).sum() + (v * R2.to(ts[0])).sum()).backward()
        return logits.detach(), v.detach(), [t.grad for t in ts]
    lr, vr, gr = run(ops_ref, [t.double() for t in [h] + prm], 'cpu')
    lg, vg, gg = run(ops, [h] + prm, 'cuda')
    torch.testing.assert_close(lg.cpu().double(), lr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(vg.cpu().double(), vr, rtol=1e-5, atol=2e-5)
    for a, b in zip(gg, gr):
        torch.testing.assert_close(a.cpu().double(), b, rtol=1e-4, atol=3e-5 * rows ** 0.5)


@pytest.mark.parametrize('N,rows,A', [(8, 4096, 4), (25, 1234, 5), (3, 1, 8), (8, 70001, 4)])
def test_a2c_loss_fused(N, rows, A):
    """One-pass A2C loss (values and d/dlogits, d/dv) == softmax / log / clip / gather / mean chain of
    policies.py:20-30 in float64, incl. probabilities below the 1e-10 clip and logits given as a column block."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N + rows + A)
    out = torch.randn(N, rows, A + 1, generator=g) * 2.0
    out[:, ::7, 0] = -60.0                                   # pi_0 ~ 1e-26 < 1e-10: clipped, gradient blocked
    act = torch.randint(0, A, (rows, N), generator=g).to(torch.uint8)
    act[::7] = 0                                             # ... and it is the action taken on some rows
    v, adv, R = (torch.randn(N, rows, generator=g) for _ in range(3))
    w = torch.rand(N, generator=g) + 0.5                     # per-agent upstream factors
    v_coef, e_coef = 0.5, 0.05

    def run(mod, dev, dt):
        o = out.to(dev, dt).requires_grad_()
        vv = v.to(dev, dt).requires_grad_()
        tot, terms = mod.a2c_loss(o[..., :A], vv, act.to(dev), adv.to(dev, dt), R.to(dev, dt), v_coef, e_coef)
        (tot * w.to(dev, dt)).sum().backward()
        return tot.detach(), terms, o.grad, vv.grad
    tr, termr, dor, dvr = run(ops_ref, 'cpu', torch.float64)
    tg, termg, dog, dvg = run(ops, 'cuda', torch.float32)
    torch.testing.assert_close(tg.cpu().double(), tr, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(termg.cpu().double(), termr, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(dog.cpu().double(), dor, rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(dvg.cpu().double(), dvr, rtol=1e-4, atol=1e-9)
    assert torch.all(dog[..., A] == 0)


def test_sample_actions_modes():
    from deeprl_network_amd import ops
    from oracle import ops_ref
    N, E, A = 8, 5000, 4
    g = torch.Generator().manual_seed(0)
    pi = torch.softmax(torch.randn(N, E, A, generator=g) * 2, -1)
    u = torch.rand(E, N, generator=g)
    for mode, kw in [(ops.SAMPLE_UNIFORM, dict(u=u)), (ops.SAMPLE_PHILOX, dict(seed=12345678901, env_id_base=777, step=4242)),
                     (ops.SAMPLE_ARGMAX, {})]:
        out = torch.zeros(E, N, dtype=torch.uint8, device='cuda')
        ref = torch.zeros(E, N, dtype=torch.uint8)
        kg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
        ops.sample_actions(pi.cuda(), out, mode, **kg)
        ops_ref.sample_actions(pi, ref, mode, **kw)
        assert torch.equal(out.cpu(), ref), mode
    # distribution sanity at scale: empirical frequencies follow pi
    E2 = 200000
    p = torch.tensor([0.1, 0.2, 0.3, 0.4]).view(1, 1, 4).expand(N, E2, 4).contiguous().cuda()
    out = torch.zeros(E2, N, dtype=torch.uint8, device='cuda')
    ops.sample_actions(p, out, ops.SAMPLE_PHILOX, seed=3, step=9)
    freq = torch.bincount(out.flatten().long(), minlength=4).float() / out.numel()
    assert torch.allclose(freq.cpu(), torch.tensor([0.1, 0.2, 0.3, 0.4]), atol=3e-3)


def test_sample_matches_numpy_choice_stream():
    """Legacy mode == the reference's np.random.choice(arange(A), p=pi) on the same global stream."""
    from deeprl_network_amd import ops
    N, A = 8, 4
    rng = np.random.RandomState(5)
    pi = rng.dirichlet(np.ones(A), size=N).astype(np.float32)
    np.random.seed(99)
    want = [np.random.choice(np.arange(A), p=p) for p in pi]
    np.random.seed(99)
    u = torch.tensor([np.random.random_sample() for _ in range(N)], dtype=torch.float32).view(1, N)
    out = torch.zeros(1, N, dtype=torch.uint8, device='cuda')
    ops.sample_actions(torch.from_numpy(pi).view(N, 1, A).cuda(), out, ops.SAMPLE_UNIFORM, u=u.cuda())
    assert out.cpu().numpy()[0].tolist() == want


@pytest.mark.parametrize('alpha', [-1.0, 0.9])
@pytest.mark.parametrize('N,E,T', [(8, 1, 60), (8, 4096, 60), (25, 100, 120)])
def test_nstep_return(alpha, N, E, T):
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(T + N)
    r = torch.randn((T, E, N) if alpha >= 0 else (T, E), generator=g)
    v = torch.randn(T, N, E, generator=g)
    done = (torch.rand(T, E, generator=g) < 0.05).to(torch.uint8)
    Rend = torch.randn(N, E, generator=g)
    idx = np.arange(N)
    dist = torch.from_numpy(np.abs(idx[:, None] - idx[None, :]).astype(np.int32))
    R1, A1 = ops.nstep_return(r.cuda(), v.cuda(), done.cuda(), Rend.cuda(), 0.99, alpha, dist.cuda())
    R2, A2 = ops_ref.nstep_return(r, v, done, Rend, 0.99, alpha, dist)
    torch.testing.assert_close(R1.cpu(), R2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(A1.cpu(), A2, rtol=1e-6, atol=1e-6)


def test_nstep_matches_reference_buffer_golden():
    """SURVEY.md 8(c) known answer of MultiAgentOnPolicyBuffer(0.99, -1)."""
    from deeprl_network_amd import ops
    r = torch.tensor([[-1.0], [-2.0], [-3.0]])
    v = torch.tensor([[[.1], [.2], [.3]]] * 3)
    done = torch.zeros(3, 1, dtype=torch.uint8)
    Rend = torch.tensor([[1.0], [2.0], [3.0]])
    R, A = ops.nstep_return(r.cuda(), v.cuda(), done.cuda(), Rend.cuda(), 0.99, -1.0, None)
    want = np.array([[-4.950001, -3.9899, -2.01], [-3.979702, -3.0098, -1.02], [-3.009403, -2.0297, -0.03]])
    np.testing.assert_allclose(R.cpu().numpy()[:, :, 0], want, rtol=1e-6)
    np.testing.assert_allclose(A.cpu().numpy()[:, :, 0], want - np.array([[.1], [.2], [.3]]), rtol=1e-5)


@pytest.mark.parametrize('G,P,max_norm', [(8, 51341, 40.0), (1, 598496, 40.0), (8, 1000, 0.5), (3, 77, -1.0)])
def test_rmsprop_tf_clip(G, P, max_norm):
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(P)
    w = torch.randn(G, P, generator=g)
    gr = torch.randn(G, P, generator=g) * 0.1
    ms = torch.ones(G, P)
    wg, msg = w.cuda(), ms.cuda()
    scratch = torch.zeros(G, 64, device='cuda')
    n1, n2 = torch.zeros(G, device='cuda'), torch.zeros(G)
    for it in range(3):
        ops.rmsprop_tf_clip(wg, gr.cuda() * (it + 1), msg, scratch, 5e-4, 0.99, 1e-5, max_norm, 0.5, n1)
        ops_ref.rmsprop_tf_clip(w, gr * (it + 1), ms, None, 5e-4, 0.99, 1e-5, max_norm, 0.5, n2)
        torch.testing.assert_close(n1.cpu(), n2, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(wg.cpu(), w, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(msg.cpu(), ms, rtol=1e-5, atol=1e-7)


def test_ops_reject_cpu_tensors():
    from deeprl_network_amd import _lib, ops
    nbr, _ = ops.neighbor_table(_masks('line'), 'cuda')
    with pytest.raises(_lib.NmarlError):
        ops.nbr_gather(torch.zeros(8, 2, 4), nbr)


def _xside_case(N, E, KX, A, m_max, seed, addends):
    H = 64
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    d = dict(h=r(N, E, H) * 0.7, c=r(N, E, H), done=(torch.rand(E, generator=g) < 0.3).float(),
             x=None if KX == 0 else r(N, E, KX) * 0.5,
             # asymmetric weights: a swapped row / column / chunk mapping of the image cannot pass
             wx=None if KX == 0 else r(N, KX, 4 * H) * 0.15 + torch.arange(4 * H).view(1, 1, -1) * 1e-3 + torch.arange(KX).view(1, -1, 1) * 1e-3,
             wh=r(N, H, 4 * H) * 0.2 + torch.arange(4 * H).view(1, 1, -1) * 1e-3, b=r(N, 4 * H) * 0.1,
             z1=r(N, E, 4 * H) if addends >= 1 else None, z2=r(N, E, 4 * H) if addends >= 2 else None,
             pi_w=r(N, H, A) * 0.5, pi_b=r(N, A) * 0.3, v_w=r(N, H + m_max * A, 1), v_b=r(N, 1))
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        others = [j for j in range(N) if j != i][:(i % m_max) + 1]
        idx[i, :len(others)] = torch.tensor(others, dtype=torch.int32)
    d['idx'] = idx
    return d


@pytest.mark.parametrize('N,E', [(8, 4096), (8, 1), (25, 130), (3, 127), (8, 257)])
@pytest.mark.parametrize('KX,addends', [(128, 0), (64, 0), (192, 1), (0, 1), (32, 2), (256, 0)])
def test_lstm_step_x_whole_preactivation_on_mfma(N, E, KX, addends):
    """nmarl_lstm_step_x: z = [x | h keep] @ [Wx; Wh] + b (+ addends) -> cell, vs the float64 restatement
    (agents/utils.py:102-113): every supported input width, ragged row counts, x as a column block of a wider buffer,
    strided sequence slots, gates output, in-place state."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    d = _xside_case(N, E, KX, 4, 2, N * 1000 + E + KX, addends)
    f64 = lambda t: None if t is None else t.double()                                    # noqa: E731
    cu = lambda t: None if t is None else t.cuda()                                       # noqa: E731
    hr, cr = torch.empty(N, E, H, dtype=torch.float64), torch.empty(N, E, H, dtype=torch.float64)
    gr = torch.empty(N, E, 4 * H, dtype=torch.float64)
    ops_ref.lstm_step_fused(f64(d['h']), f64(d['wh']), f64(d['b']), f64(d['z1']), f64(d['z2']), f64(d['c']), f64(d['done']),
                            gr, cr, hr, xs=(f64(d['x']), f64(d['wx']), None))
    img = ops.lstm_wimage(cu(d['wx']), cu(d['wh']))
    assert img.shape == (N, (KX + 64) * 320)
    xg = None
    if KX:
        wide = torch.zeros(N, E, KX + 64, device='cuda')          # x = columns [32, 32 + KX) of a wider buffer
        xg = wide[:, :, 32:32 + KX]
        xg.copy_(d['x'])
    Hbuf = torch.zeros(N, 3, E, H, device='cuda'); Cbuf = torch.zeros(N, 3, E, H, device='cuda')
    Gbuf = torch.zeros(N, 3, E, 4 * H, device='cuda')
    Hbuf[:, 0].copy_(d['h']); Cbuf[:, 0].copy_(d['c'])
    ops.lstm_step_fused(Hbuf[:, 0], None, cu(d['b']), cu(d['z1']), cu(d['z2']), Cbuf[:, 0], cu(d['done']), Gbuf[:, 1],
                        Cbuf[:, 1], Hbuf[:, 1], xs=(xg, None, img))
    tol = dict(rtol=3e-5, atol=5e-6)
    torch.testing.assert_close(Hbuf[:, 1].cpu().double(), hr, **tol)
    torch.testing.assert_close(Cbuf[:, 1].cpu().double(), cr, **tol)
    torch.testing.assert_close(Gbuf[:, 1].cpu().double(), gr, **tol)
    assert torch.all(Hbuf[:, 2] == 0) and torch.all(Gbuf[:, 2] == 0) and torch.all(Gbuf[:, 0] == 0)
    hg, cg = cu(d['h']), cu(d['c'])                                # in place, no gates (rollout)
    ops.lstm_step_fused(hg, None, cu(d['b']), cu(d['z1']), cu(d['z2']), cg, cu(d['done']), None, cg, hg, xs=(xg, None, img))
    torch.testing.assert_close(hg.cpu().double(), hr, **tol)
    torch.testing.assert_close(cg.cpu().double(), cr, **tol)


@pytest.mark.parametrize('N,E,A,m_max', [(8, 4096, 4, 2), (25, 130, 5, 4), (3, 127, 8, 2), (8, 1, 4, 2)])
@pytest.mark.parametrize('KX,addends', [(128, 0), (64, 1), (192, 0)])
@pytest.mark.parametrize('mode', [1, 2])
def test_lstm_step_x_heads(N, E, A, m_max, KX, addends, mode):
    """The head epilogues on top of the x-side step: forward('p') + draw (kind 1), forward('v') (kind 2) and both in
    one kernel (kind 3, quirk Q1: the value re-step re-uses the x-side part and the resident Wh chunks)."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    d = _xside_case(N, E, KX, A, m_max, N * 77 + E + A + KX, addends)
    f64 = lambda t: None if t is None else t.double()                                    # noqa: E731
    cu = lambda t: None if t is None else t.cuda()                                       # noqa: E731
    draw = dict(mode=mode, seed=5, env_id_base=40, step=3)
    img = ops.lstm_wimage(cu(d['wx']), cu(d['wh']))
    xs_g = (cu(d['x']), None, img)
    xs_r = (f64(d['x']), f64(d['wx']), None)
    # oracle: policy half (float64), draw from the KERNEL's probabilities, then the value half
    hr, cr = d['h'].double(), d['c'].double()
    pir, actr = torch.zeros(N, E, A, dtype=torch.float64), torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.lstm_step_policy(hr, f64(d['wh']), f64(d['b']), f64(d['z1']), f64(d['z2']), cr, f64(d['done']), cr, hr,
                             f64(d['pi_w']), f64(d['pi_b']), pir, actr, xs=xs_r, **draw)
    # kind 3
    hg, cg = cu(d['h']), cu(d['c'])
    pig, actg = torch.zeros(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda')
    vg = torch.zeros(N, E, device='cuda')
    ops.lstm_step_policy_value(hg, None, cu(d['b']), cu(d['z1']), cu(d['z2']), cg, cu(d['done']), cu(d['pi_w']), cu(d['pi_b']),
                               pig, actg, cu(d['v_w']), cu(d['v_b']), cu(d['idx']), A, vg, xs=xs_g, **draw)
    tol = dict(rtol=3e-5, atol=5e-6)
    torch.testing.assert_close(hg.cpu().double(), hr, **tol)
    torch.testing.assert_close(cg.cpu().double(), cr, **tol)
    torch.testing.assert_close(pig.cpu().double(), pir, **tol)
    act_chk = torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.sample_actions(pig.cpu(), act_chk, **draw)
    assert torch.equal(actg.cpu(), act_chk)
    vr = torch.zeros(N, E, dtype=torch.float64)
    ops_ref.lstm_step_value(hr, f64(d['wh']), f64(d['b']), f64(d['z1']), f64(d['z2']), cr, f64(d['done']), torch.empty_like(cr),
                            torch.empty_like(hr), f64(d['v_w']), f64(d['v_b']), act_chk, d['idx'], A, vr, xs=xs_r)
    torch.testing.assert_close(vg.cpu().double(), vr, rtol=1e-4, atol=3e-5)
    # kinds 1 and 2 as separate launches give the same numbers
    h1, c1 = cu(d['h']), cu(d['c'])
    pi1, act1 = torch.zeros(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda')
    ops.lstm_step_policy(h1, None, cu(d['b']), cu(d['z1']), cu(d['z2']), c1, cu(d['done']), c1, h1, cu(d['pi_w']), cu(d['pi_b']),
                         pi1, act1, xs=xs_g, **draw)
    assert torch.equal(act1, actg)
    torch.testing.assert_close(pi1, pig, rtol=1e-6, atol=1e-7)
    h2, c2, v2 = torch.zeros_like(h1), torch.zeros_like(c1), torch.zeros(N, E, device='cuda')
    ops.lstm_step_value(h1, None, cu(d['b']), cu(d['z1']), cu(d['z2']), c1, cu(d['done']), c2, h2, cu(d['v_w']), cu(d['v_b']),
                        act1, cu(d['idx']), A, v2, xs=xs_g)
    torch.testing.assert_close(v2, vg, rtol=1e-5, atol=1e-6)


def test_lstm_sequence_x_fwd_bwd():
    """The update's recurrence with the x-side product inside the step (ops.lstm_sequence_x) vs plain autograd over the
    restatement: outputs and all gradients (s, wx, wh, b, h0, c0)."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    N, T, E, KX, H = 3, 5, 130, 128, 64
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    s, wx, wh, b = r(N, T, E, KX) * 0.5, r(N, KX, 4 * H) * 0.15, r(N, H, 4 * H) * 0.2, r(N, 4 * H) * 0.1
    h0, c0 = r(N, E, H) * 0.5, r(N, E, H) * 0.5
    done = torch.zeros(T, E)
    done[0] = (torch.rand(E, generator=g) < 0.4).float()
    dH = r(N, T, E, H)
    ref_in = [t.double().requires_grad_(True) for t in (s, wx, wh, b, h0, c0)]
    out_r = ops_ref.lstm_sequence_x(*ref_in, done.double(), (0,), None)
    (out_r * dH.double()).sum().backward()
    gpu_in = [t.cuda().requires_grad_(True) for t in (s, wx, wh, b, h0, c0)]
    img = ops.lstm_wimage(gpu_in[1].detach(), gpu_in[2].detach())
    out_g = ops.lstm_sequence_x(*gpu_in, done.cuda(), (0,), img)
    (out_g * dH.cuda()).sum().backward()
    torch.testing.assert_close(out_g.detach().cpu().double(), out_r.detach(), rtol=1e-4, atol=1e-5)
    for name, a, bb in zip('s wx wh b h0 c0'.split(), gpu_in, ref_in):
        torch.testing.assert_close(a.grad.cpu().double(), bb.grad, rtol=2e-3, atol=2e-4, msg=lambda m, n=name: '%s: %s' % (n, m))


@pytest.mark.parametrize('N,E', [(8, 4096), (8, 1), (25, 130), (3, 127)])
@pytest.mark.parametrize('KM,masked,with_rec', [(0, True, True), (0, False, False), (64, True, True), (64, False, True)])
def test_lstm_bptt_step_fused(N, E, KM, masked, with_rec):
    """nmarl_lstm_bptt_step (cell backward + [dx | dh] = dz @ [wxm; wh]^T on MFMA, relu / done masks) vs the float64
    restatement: strided sequence slots, ragged rows, absent recurrent / cell gradients."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    g = torch.Generator().manual_seed(N * 31 + E + KM)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    gates = torch.cat([torch.sigmoid(r(N, E, 3 * H)), torch.tanh(r(N, E, H))], dim=-1)
    c_prev, c_new = r(N, E, H), r(N, E, H) * 0.8
    done = (torch.rand(E, generator=g) < 0.3).float()
    dh, dh2, dc = r(N, E, H), (r(N, E, H) if with_rec else None), (r(N, E, H) if with_rec else None)
    wh = r(N, H, 4 * H) * 0.2 + torch.arange(4 * H).view(1, 1, -1) * 1e-3 + torch.arange(H).view(1, -1, 1) * 2e-3
    wxm = None if KM == 0 else r(N, KM, 4 * H) * 0.2 + torch.arange(4 * H).view(1, 1, -1) * 1e-3
    hm = torch.relu(r(N, E, 3 * H))                       # the mask: last third of a [N,E,3H] LSTM input (row pitch 192)
    f64 = lambda t: None if t is None else t.double()                                    # noqa: E731
    cu = lambda t: None if t is None else t.cuda()                                       # noqa: E731
    dz_r, dcp_r, dhd_r = torch.empty(N, E, 4 * H, dtype=torch.float64), torch.empty(N, E, H, dtype=torch.float64), torch.empty(N, E, H, dtype=torch.float64)
    dx_r = torch.empty(N, E, H, dtype=torch.float64) if KM else None
    mask_r = f64(hm[:, :, 2 * H:]) if (KM and masked) else None
    ops_ref.bptt_step(f64(gates), f64(c_prev), f64(c_new), f64(done), f64(dh), f64(dh2), f64(dc), (f64(wxm), f64(wh), None), dz_r,
                      dcp_r, dhd_r, masked, dx=dx_r, mask=mask_r)
    ws = (cu(wxm), cu(wh), ops.lstm_bptt_wimage(cu(wxm), cu(wh)))
    G = torch.zeros(N, 3, E, 4 * H, device='cuda'); G[:, 1].copy_(gates)
    C = torch.zeros(N, 3, E, H, device='cuda'); C[:, 1].copy_(c_prev); C[:, 2].copy_(c_new)
    dZ = torch.zeros(N, 3, E, 4 * H, device='cuda')
    dcp, dhd = torch.zeros(N, E, H, device='cuda'), torch.zeros(N, E, H, device='cuda')
    dxg = torch.zeros(N, 2, E, H, device='cuda') if KM else None
    hmg = cu(hm)
    ops.bptt_step(G[:, 1], C[:, 1], C[:, 2], cu(done), cu(dh), cu(dh2), cu(dc), ws, dZ[:, 1], dcp, dhd, masked,
                  dx=None if dxg is None else dxg[:, 1], mask=hmg[:, :, 2 * H:] if (KM and masked) else None)
    tol = dict(rtol=2e-5, atol=3e-6)
    torch.testing.assert_close(dZ[:, 1].cpu().double(), dz_r, **tol)
    torch.testing.assert_close(dcp.cpu().double(), dcp_r, **tol)
    torch.testing.assert_close(dhd.cpu().double(), dhd_r, rtol=1e-4, atol=2e-5)
    assert torch.all(dZ[:, 0] == 0) and torch.all(dZ[:, 2] == 0)
    if KM:
        torch.testing.assert_close(dxg[:, 1].cpu().double(), dx_r, rtol=1e-4, atol=2e-5)
        assert torch.all(dxg[:, 0] == 0)
    # the same step with the bias gradient accumulated on the way: same outputs bit for bit, db_part += column sums of dz
    part = ops.bptt_step_db_parts(N, E, H, 'cuda')
    dZ2, dcp2, dhd2 = torch.zeros_like(dZ), torch.zeros_like(dcp), torch.zeros_like(dhd)
    for rep in range(2):
        ops.bptt_step(G[:, 1], C[:, 1], C[:, 2], cu(done), cu(dh), cu(dh2), cu(dc), ws, dZ2[:, 1], dcp2, dhd2, masked,
                      dx=None if dxg is None else dxg[:, 1], mask=hmg[:, :, 2 * H:] if (KM and masked) else None, db_part=part)
    assert torch.equal(dZ2, dZ) and torch.equal(dcp2, dcp) and torch.equal(dhd2, dhd)
    torch.testing.assert_close(part.sum(1).cpu().double(), 2 * dz_r.sum(1), rtol=1e-4, atol=2e-5 * max(1.0, E ** 0.5))


def _forward_cells(gates, c0, done):
    """The cell-state sequence a forward pass would have saved with these gates: c[t + 1] = gf * (c[t] * keep_t) + gi * gu in
    float32, operation for operation what the forward kernels compute (csrc/lstm_mfma.hip, a2c.hip).  The one-launch BPTT
    kernels RECOMPUTE c_t from the gates instead of reading it, so their inputs must be a consistent forward trace."""
    N, T, E, H4 = gates.shape
    H = H4 // 4
    call = torch.empty(N, T + 1, E, H)
    call[:, 0] = c0
    for t in range(T):
        keep = (1.0 - done[t]).view(1, E, 1)
        call[:, t + 1] = gates[:, t, :, H:2 * H] * (call[:, t] * keep) + gates[:, t, :, :H] * gates[:, t, :, 3 * H:]
    return call


@pytest.mark.parametrize('N,T,E', [(8, 12, 4096), (8, 60, 256), (3, 5, 127), (25, 4, 130), (2, 1, 1)])
def test_lstm_bptt_seq_one_launch(N, T, E):
    """nmarl_lstm_bptt_seq (the whole reverse recurrence in one launch, state on chip) vs T launches of
    nmarl_lstm_bptt_step (bit for bit: same arithmetic) and vs the float64 restatement; bias-gradient partials;
    strided sequence buffers (slots of wider allocations), ragged rows, dones inside the sequence."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    g = torch.Generator().manual_seed(N * 131 + T * 7 + E)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    gates = torch.cat([torch.sigmoid(r(N, T, E, 3 * H)), torch.tanh(r(N, T, E, H))], dim=-1)
    c0 = r(N, E, H) * 0.8
    done = (torch.rand(T, E, generator=g) < 0.2).float()
    call = _forward_cells(gates, c0, done)
    dhs = r(N, T, E, H)
    wh = r(N, H, 4 * H) * 0.1 + torch.arange(4 * H).view(1, 1, -1) * 1e-4 + torch.arange(H).view(1, -1, 1) * 2e-4
    dz_r = torch.empty(N, T, E, 4 * H, dtype=torch.float64)
    db_r, dh0_r, dc0_r = ops_ref.bptt_seq(gates.double(), call.double(), done.double(), dhs.double(), None, dz_r,
                                          want_state_grad=True, wh=wh.double())
    # device buffers: slots of wider allocations (agent stride > T * E * W)
    G = torch.zeros(N, T + 2, E, 4 * H, device='cuda'); G[:, 1:T + 1].copy_(gates)
    C = torch.zeros(N, T + 3, E, H, device='cuda'); C[:, 1:T + 2].copy_(call)
    D = torch.zeros(N, T + 1, E, H, device='cuda'); D[:, :T].copy_(dhs)
    dZ = torch.zeros(N, T + 2, E, 4 * H, device='cuda')
    whg, doneg = wh.cuda(), done.cuda()
    img = ops.lstm_bptt_wimage(None, whg)
    db, dh0, dc0 = ops.bptt_seq(G[:, 1:T + 1], C[:, 1:T + 2], doneg, D[:, :T], img, dZ[:, 1:T + 1], want_state_grad=True)
    assert torch.all(dZ[:, 0] == 0) and torch.all(dZ[:, T + 1] == 0)
    # the step-by-step path
    dZ2 = torch.zeros(N, T, E, 4 * H, device='cuda')
    dc = torch.zeros(N, E, H, device='cuda'); dcn = torch.empty_like(dc)
    dh_a, dh_b, dh_rec = torch.empty_like(dc), torch.empty_like(dc), None
    ws = (None, whg, img)
    for t in range(T - 1, -1, -1):
        ops.bptt_step(G[:, 1 + t], C[:, 1 + t], C[:, 2 + t], doneg[t], D[:, t], dh_rec, dc, ws, dZ2[:, t], dcn, dh_a, True)
        dc, dcn = dcn, dc
        dh_rec, dh_a, dh_b = dh_a, dh_b, dh_a
    assert torch.equal(dZ[:, 1:T + 1], dZ2)
    assert torch.equal(dh0, dh_rec) and torch.equal(dc0, dc)
    torch.testing.assert_close(dZ[:, 1:T + 1].cpu().double(), dz_r, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dh0.cpu().double(), dh0_r, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dc0.cpu().double(), dc0_r, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(db.cpu().double(), db_r, rtol=1e-4, atol=1e-4 * max(1.0, float(db_r.abs().max())))
    torch.testing.assert_close(db.cpu().double(), dZ2.double().sum(dim=(1, 2)).cpu(), rtol=1e-5,
                               atol=1e-6 * max(1.0, float(db_r.abs().max())) * (T * E) ** 0.5)


@pytest.mark.parametrize('N,T,E,O', [(8, 12, 4096, 5), (3, 5, 127, 6), (25, 4, 130, 6), (2, 1, 1, 5), (4, 7, 33, 3), (2, 3, 70, 8)])
def test_lstm_bptt_seq_expands_the_heads_gradient_itself(N, T, E, O):
    """nmarl_lstm_bptt_seq_dy (round 6): the heads' dL/dh handed over as dy8 [N,T*E,8] = [d logits | d v | 0] + the heads' weights
    hw [N,64,O] and formed inside the launch (two more k-steps of every step's transposed product) against nmarl_lstm_bptt_seq on
    the tensor dL/dh = dy hw^T (float64 product, rounded once): dz, the bias gradient and the initial state's gradient at fp32
    summation-order tolerance (the fp32 matrix-core accumulation of 5-8 terms against a rounded float64 sum); dones inside the
    sequence, ragged rows, strided sequence buffers."""
    from deeprl_network_amd import ops
    H = 64
    g = torch.Generator().manual_seed(N * 17 + T * 3 + E + O)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    gates = torch.cat([torch.sigmoid(r(N, T, E, 3 * H)), torch.tanh(r(N, T, E, H))], dim=-1)
    done = (torch.rand(T, E, generator=g) < 0.2).float()
    call = _forward_cells(gates, r(N, E, H) * 0.8, done)
    dy8 = torch.zeros(N, T * E, 8)
    dy8[:, :, :O] = r(N, T * E, O)
    hw = r(N, H, O) * 0.3
    dhs = torch.bmm(dy8[:, :, :O].double(), hw.double().transpose(1, 2)).float().view(N, T, E, H)
    wh = r(N, H, 4 * H) * 0.1
    G = torch.zeros(N, T + 2, E, 4 * H, device='cuda'); G[:, 1:T + 1].copy_(gates)
    C = torch.zeros(N, T + 3, E, H, device='cuda'); C[:, 1:T + 2].copy_(call)
    img = ops.lstm_bptt_wimage(None, wh.cuda())
    out = []
    for head_dy in (None, (dy8.cuda(), hw.cuda())):
        dZ = torch.zeros(N, T + 2, E, 4 * H, device='cuda')
        db, dh0, dc0 = ops.bptt_seq(G[:, 1:T + 1], C[:, 1:T + 2], done.cuda(), dhs.cuda() if head_dy is None else None, img, dZ[:, 1:T + 1],
                                    want_state_grad=True, head_dy=head_dy)
        assert torch.all(dZ[:, 0] == 0) and torch.all(dZ[:, T + 1] == 0)
        out.append((dZ[:, 1:T + 1].clone(), db, dh0, dc0))
    for a, b in zip(*out):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6 * float(b.abs().max()))


@pytest.mark.parametrize('N,rows,A,m_max', [(8, 4096, 4, 2), (25, 1003, 5, 4), (3, 1, 4, 2), (8, 70001, 4, 2), (5, 129, 3, 2), (2, 300, 7, 1)])
@pytest.mark.parametrize('want_dh', [True, False])
def test_heads_loss_one_pass_vs_torch(N, rows, A, m_max, want_dh):
    """nmarl_heads_loss (round 6): the update's actor / critic heads (policies.py:50-77), the A2C loss (policies.py:20-30: softmax,
    log of the clamped probabilities, entropy, value loss with the neighbour-action term) and the heads' backward in one pass over
    h, against float64 torch autograd on the reference's formulas: the three loss terms, d logits | d v (dy8), dL/dh and the
    gradients of pi_w, pi_b, v_w (h part and neighbour-action part), v_b.  A = 4 / 5: the compile-time forms; 3 / 7: the generic one."""
    from deeprl_network_amd import ops
    H = 64
    g = torch.Generator().manual_seed(N * 5 + rows + A)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    h = torch.tanh(r(N, rows, H))
    nbr = torch.tensor([[(i + k + 1) % N if k < 1 + (i % m_max) else -1 for k in range(m_max)] for i in range(N)], dtype=torch.int32)
    pi_w, pi_b, v_w, v_b = r(N, H, A) * .3, r(N, A) * .1, r(N, H + m_max * A, 1) * .2, r(N, 1) * .1
    action = torch.randint(0, A, (rows, N), generator=g).to(torch.uint8)
    adv, R = r(N, rows), r(N, rows)
    v_coef, e_coef = 0.5, 0.01
    # float64 reference (agents/models.py `_loss` fallback = the reference's prepare_loss)
    P = [t.double().requires_grad_(True) for t in (h, pi_w, pi_b, v_w, v_b)]
    hd, pw, pb, vw, vb = P
    logits = torch.baddbmm(pb.unsqueeze(1), hd, pw)
    na = torch.zeros(N, rows, m_max * A, dtype=torch.float64)
    for i in range(N):
        for k in range(m_max):
            j = int(nbr[i, k])
            if j >= 0:
                na[i, torch.arange(rows), k * A + action[:, j].long()] = 1.0
    v = (torch.baddbmm(vb.unsqueeze(1), hd, vw[:, :H]) + torch.bmm(na, vw[:, H:])).squeeze(-1)
    pi = torch.softmax(logits, dim=-1)
    log_pi = torch.log(torch.clamp(pi, 1e-10, 1.0))
    ent = -(pi * log_pi).sum(-1)
    logp_a = log_pi.gather(-1, action.t().long().unsqueeze(-1)).squeeze(-1)
    terms_r = torch.stack([-(logp_a * adv.double()).mean(-1), (R.double() - v).pow(2).mean(-1) * 0.5 * v_coef, -ent.mean(-1) * e_coef], dim=1)
    logits.retain_grad(); v.retain_grad()
    terms_r.sum().backward()
    c = lambda t: t.cuda()                                                              # noqa: E731
    out = ops.heads_loss(c(h), c(pi_w), c(pi_b), c(v_w), c(v_b), c(action), c(nbr), A, c(adv), c(R), v_coef, e_coef, want_dh=want_dh)
    tol = dict(rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(out['terms'].cpu().double(), terms_r.detach(), rtol=2e-5, atol=1e-7)
    dy8 = out['dy8'].cpu().double()
    scale = float(logits.grad.abs().max())
    torch.testing.assert_close(dy8[:, :, :A], logits.grad, rtol=2e-4, atol=2e-6 * scale)
    torch.testing.assert_close(dy8[:, :, A], v.grad, rtol=2e-4, atol=2e-6 * float(v.grad.abs().max()))
    assert float(dy8[:, :, A + 1:].abs().max() if A + 1 < 8 else 0.0) == 0.0
    assert (out['dh'] is not None) == want_dh
    if want_dh:
        torch.testing.assert_close(out['dh'].cpu().double(), hd.grad, rtol=2e-4, atol=2e-6 * float(hd.grad.abs().max()))
    for key, ref in (('pi_w', pw.grad), ('pi_b', pb.grad), ('v_w', vw.grad), ('v_b', vb.grad)):
        torch.testing.assert_close(out[key].cpu().double().reshape(ref.shape), ref, rtol=2e-4, atol=3e-6 * float(ref.abs().max()) + 1e-9), key


@pytest.mark.parametrize('N,E,A,m_max', [(8, 4096, 4, 2), (25, 130, 5, 4), (5, 127, 4, 2), (25, 1024, 5, 4)])
@pytest.mark.parametrize('kind', [1, 2])
def test_lstm_step_x_in_kernel_message_term(N, E, A, m_max, kind):
    """The message term of a coupled net computed by the step kernel's pre-phase (nmarl_lstm_step_x_msg) vs the float64
    restatement: lstm_comm hm = relu(gather(h) W_msg + b) (kind 1, K = 64 m_max <= 128) / lstm_ic3 s = mean(h) W_msg + b +
    enc (kind 2), ragged -1 padded neighbour tables, message output stored into a column block, policy and value heads."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    if kind == 1 and H * m_max > ops.MSG_MAX_K:
        pytest.skip('message input wider than the pre-phase supports (falls back to separate launches)')
    g = torch.Generator().manual_seed(N * 13 + E + kind)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    KXg = 2 * H if kind == 1 else 0
    KX = KXg + H
    Km = H * m_max if kind == 1 else H
    h, c, done = r(N, E, H) * 0.7, r(N, E, H), (torch.rand(E, generator=g) < 0.3).float()
    xg = torch.relu(r(N, E, KXg)) if KXg else None
    enc = torch.tanh(r(N, E, H)) if kind == 2 else None
    wx = r(N, KX, 4 * H) * 0.15 + torch.arange(4 * H).view(1, 1, -1) * 1e-3 + torch.arange(KX).view(1, -1, 1) * 1e-3
    wh, b = r(N, H, 4 * H) * 0.2, r(N, 4 * H) * 0.1
    w_msg = r(N, Km, H) * 0.2 + torch.arange(H).view(1, 1, -1) * 2e-3 + torch.arange(Km).view(1, -1, 1) * 1e-3
    b_msg = r(N, H) * 0.2
    pi_w, pi_b, v_w, v_b = r(N, H, A) * 0.5, r(N, A) * 0.3, r(N, H + m_max * A, 1), r(N, 1)
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        others = [j for j in range(N) if j != i][:(i % m_max) + (0 if i == N - 1 else 1)]   # the last agent: no neighbours
        if others:
            idx[i, :len(others)] = torch.tensor(others, dtype=torch.int32)
    f64 = lambda t: None if t is None else t.double()                                    # noqa: E731
    cu = lambda t: None if t is None else t.cuda()                                       # noqa: E731
    draw = dict(mode=2, seed=5, env_id_base=40, step=3)
    # oracle
    out_r = torch.zeros(N, E, H, dtype=torch.float64)
    msg_r = dict(kind=kind, nbr_idx=idx, w_msg=f64(w_msg), b_msg=f64(b_msg), enc=f64(enc), out=out_r)
    mm_r, mm_g = torch.zeros(N, E, H, dtype=torch.float64), torch.zeros(N, 2, E, H, device='cuda')
    if kind == 2:                            # lstm_ic3: the policy step also keeps mean_j(h_j), the message layer's input
        msg_r['mean_out'] = mm_r
    hr, cr = torch.empty(N, E, H, dtype=torch.float64), torch.empty(N, E, H, dtype=torch.float64)
    pir, actr = torch.zeros(N, E, A, dtype=torch.float64), torch.zeros(E, N, dtype=torch.uint8)
    gr = torch.zeros(N, E, 4 * H, dtype=torch.float64)
    ops_ref.lstm_step_policy(f64(h), f64(wh), f64(b), None, None, f64(c), f64(done), cr, hr, f64(pi_w), f64(pi_b), pir, actr,
                             xs=(f64(xg), f64(wx), None, None, msg_r), gates=gr, **draw)
    # product: message output into the last third of a [N,E,KX] slot (kind 1) / a separate slot (kind 2)
    img, mimg = ops.lstm_wimage(cu(wx), cu(wh)), ops.lstm_msg_wimage(cu(w_msg))
    slot = torch.zeros(N, E, KX, device='cuda')
    if KXg:
        slot[:, :, :KXg].copy_(xg)
    msg_g = dict(kind=kind, nbr_idx=cu(idx), w_msg=cu(w_msg), b_msg=cu(b_msg), img=mimg, enc=cu(enc), out=slot[:, :, KXg:])
    if kind == 2:
        msg_g['mean_out'] = mm_g[:, 1]
    hg, cg = torch.zeros(N, E, H, device='cuda'), torch.zeros(N, E, H, device='cuda')
    pig, actg, gg = torch.zeros(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda'), torch.zeros(N, E, 4 * H, device='cuda')
    ops.lstm_step_policy(cu(h), None, cu(b), None, None, cu(c), cu(done), cg, hg, cu(pi_w), cu(pi_b), pig, actg,
                         xs=(slot[:, :, :KXg] if KXg else None, None, img, None, msg_g), gates=gg, **draw)
    tol = dict(rtol=5e-5, atol=1e-5)
    torch.testing.assert_close(slot[:, :, KXg:].cpu().double(), out_r, **tol)
    if kind == 2:
        torch.testing.assert_close(mm_g[:, 1].cpu().double(), mm_r, rtol=1e-6, atol=1e-6)
        assert float(mm_r.abs().max()) > 0 and float(mm_g[:, 0].abs().max()) == 0.0
        msg_g.pop('mean_out'), msg_r.pop('mean_out')
    torch.testing.assert_close(hg.cpu().double(), hr, **tol)
    torch.testing.assert_close(cg.cpu().double(), cr, **tol)
    torch.testing.assert_close(gg.cpu().double(), gr, **tol)
    torch.testing.assert_close(pig.cpu().double(), pir, **tol)
    # value step from the new state: message recomputed from h', not stored
    act_chk = torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.sample_actions(pig.cpu(), act_chk, **draw)
    assert torch.equal(actg.cpu(), act_chk)
    vr = torch.zeros(N, E, dtype=torch.float64)
    msg_r2 = dict(msg_r, out=None)
    ops_ref.lstm_step_value(hr, f64(wh), f64(b), None, None, cr, f64(done), torch.empty_like(cr), torch.empty_like(hr), f64(v_w),
                            f64(v_b), act_chk, idx, A, vr, xs=(f64(xg), f64(wx), None, None, msg_r2))
    vg, h2, c2 = torch.zeros(N, E, device='cuda'), torch.zeros_like(hg), torch.zeros_like(cg)
    keep = slot.clone()
    ops.lstm_step_value(hg, None, cu(b), None, None, cg, cu(done), c2, h2, cu(v_w), cu(v_b), actg, cu(idx), A, vg,
                        xs=(slot[:, :, :KXg] if KXg else None, None, img, None, dict(msg_g, out=None)))
    torch.testing.assert_close(vg.cpu().double(), vr, rtol=2e-4, atol=5e-5)
    assert torch.equal(slot, keep)
    # both steps in ONE launch (head kind 3 + message term: the blocks hand their new h over inside the launch): the policy half
    # bit-equal to the separate policy step, the value equal to the separate value step up to the order of its sums; repeated
    # calls on the same flag words (generations)
    if not ops.step_handoff_supported(N, E, 'cuda'):
        return
    sync = ops.step_sync_words(N, E, 'cuda')
    for rep in range(3):
        slot1 = torch.zeros(N, E, KX, device='cuda')
        if KXg:
            slot1[:, :, :KXg].copy_(xg)
        h1, c1, g1 = torch.zeros_like(hg), torch.zeros_like(cg), torch.zeros_like(gg)
        pi1, act1, v1 = torch.zeros_like(pig), torch.zeros_like(actg), torch.zeros(N, E, device='cuda')
        msg1 = dict(msg_g, out=slot1[:, :, KXg:], sync=sync)
        ops.lstm_step_policy_value(cu(h), None, cu(b), None, None, cu(c), cu(done), cu(pi_w), cu(pi_b), pi1, act1, cu(v_w), cu(v_b),
                                   cu(idx), A, v1, xs=(slot1[:, :, :KXg] if KXg else None, None, img, None, msg1), h_out=h1, c_out=c1,
                                   gates=g1, **draw)
        ops.check_coupled_status()
        assert torch.equal(h1, hg) and torch.equal(c1, cg) and torch.equal(g1, gg), rep
        assert torch.equal(pi1, pig) and torch.equal(act1, actg) and torch.equal(slot1, slot), rep
        torch.testing.assert_close(v1, vg, rtol=2e-5, atol=5e-6)
        torch.testing.assert_close(v1.cpu().double(), vr, rtol=2e-4, atol=5e-5)
    assert int(sync[0].item()) == 3 and int(sync[1].item()) == 0 and int(sync[2].item()) == 0
    if kind != 2:
        return
    # lstm_ic3: the observation encoder enc = tanh([x_i | x_nbr] W_ob + b_ob) inside the same launch, from the compact observation
    # (own features only, F = 12: 16-byte pieces, F (1 + m_max) <= 64 inputs), its output written to the enc slot
    Fo = 12
    nbr_self = torch.cat([torch.arange(N, dtype=torch.int32).view(-1, 1), idx], dim=1)
    xo = r(E, N, Fo)
    w_ob, b_ob = r(N, Fo * (1 + m_max), H) * 0.3, r(N, H) * 0.2
    pad = torch.zeros(N, 64, H, device='cuda')
    oimg = ops.lstm_ob_wimage(cu(w_ob), pad)
    enc_r = torch.zeros(N, E, H, dtype=torch.float64)
    msg_ro = dict(msg_r, enc=enc_r, out=torch.zeros(N, E, H, dtype=torch.float64),
                  ob=dict(x=f64(xo), nbr=nbr_self, w=f64(w_ob), b=f64(b_ob)))
    h_r, c_r, g_r = torch.empty_like(hr), torch.empty_like(cr), torch.zeros_like(gr)
    pi_r, act_r, v_r = torch.zeros_like(pir), torch.zeros_like(actr), torch.zeros(N, E, dtype=torch.float64)
    ops_ref.lstm_step_policy_value(f64(h), f64(wh), f64(b), None, None, f64(c), f64(done), f64(pi_w), f64(pi_b), pi_r, act_r, f64(v_w),
                                   f64(v_b), idx, A, v_r, xs=(None, f64(wx), None, None, msg_ro), h_out=h_r, c_out=c_r, gates=g_r,
                                   defer_action_term=True, **draw)
    enc_g, s_g = torch.full((N, E, H), 7.0, device='cuda'), torch.zeros(N, E, H, device='cuda')
    h1, c1, g1 = torch.zeros_like(hg), torch.zeros_like(cg), torch.zeros_like(gg)
    pi1, act1, v1 = torch.zeros_like(pig), torch.zeros_like(actg), torch.zeros(N, E, device='cuda')
    msg_o = dict(msg_g, enc=enc_g, out=s_g, sync=sync, ob=dict(x=cu(xo), nbr=cu(nbr_self), img=oimg, b=cu(b_ob)))
    ops.lstm_step_policy_value(cu(h), None, cu(b), None, None, cu(c), cu(done), cu(pi_w), cu(pi_b), pi1, act1, cu(v_w), cu(v_b),
                               cu(idx), A, v1, xs=(None, None, img, None, msg_o), h_out=h1, c_out=c1, gates=g1,
                               defer_action_term=True, **draw)
    ops.check_coupled_status()
    torch.testing.assert_close(enc_g.cpu().double(), enc_r, rtol=2e-5, atol=5e-6)
    torch.testing.assert_close(s_g.cpu().double(), msg_ro['out'], **tol)
    torch.testing.assert_close(h1.cpu().double(), h_r, **tol)
    torch.testing.assert_close(c1.cpu().double(), c_r, **tol)
    torch.testing.assert_close(g1.cpu().double(), g_r, **tol)
    torch.testing.assert_close(pi1.cpu().double(), pi_r, **tol)
    torch.testing.assert_close(v1.cpu().double(), v_r, rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize('N,E,A,m_max', [(8, 4096, 4, 2), (5, 127, 4, 2), (6, 300, 5, 1)])
def test_lstm_step_x_dial_message_term(N, E, A, m_max):
    """lstm_dial's receiver side inside the step kernel (nmarl_lstm_step_x_msg kind 3, agents/utils.py:560-580) vs the float64
    restatement: hm = relu(gather(msg) W_msg + b) from the SENDERS' message vectors, s = hm + enc the LSTM input; hm and s
    stored (the update's relu mask / saved input); policy step (also IN PLACE: the pre-phase reads no h) and value step."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    g = torch.Generator().manual_seed(N * 17 + E)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    Km = H * m_max
    h, c, done = r(N, E, H) * 0.7, r(N, E, H), (torch.rand(E, generator=g) < 0.3).float()
    src, enc = torch.relu(r(N, E, H)), torch.relu(r(N, E, H))
    wx = r(N, H, 4 * H) * 0.15 + torch.arange(4 * H).view(1, 1, -1) * 1e-3 + torch.arange(H).view(1, -1, 1) * 1e-3
    wh, b = r(N, H, 4 * H) * 0.2, r(N, 4 * H) * 0.1
    w_msg = r(N, Km, H) * 0.2 + torch.arange(H).view(1, 1, -1) * 2e-3 + torch.arange(Km).view(1, -1, 1) * 1e-3
    b_msg = r(N, H) * 0.2
    pi_w, pi_b, v_w, v_b = r(N, H, A) * 0.5, r(N, A) * 0.3, r(N, H + m_max * A, 1), r(N, 1)
    idx = -torch.ones(N, m_max, dtype=torch.int32)
    for i in range(N):
        others = [j for j in range(N) if j != i][:(i % m_max) + (0 if i == N - 1 else 1)]   # the last agent: no neighbours
        if others:
            idx[i, :len(others)] = torch.tensor(others, dtype=torch.int32)
    f64 = lambda t: None if t is None else t.double()                                    # noqa: E731
    cu = lambda t: None if t is None else t.cuda()                                       # noqa: E731
    draw = dict(mode=2, seed=5, env_id_base=40, step=3)
    s_r, hm_r = torch.zeros(N, E, H, dtype=torch.float64), torch.zeros(N, E, H, dtype=torch.float64)
    mfc_w, mfc_b = r(N, H, H) * 0.3, r(N, H) * 0.2
    nxt_r = torch.zeros(N, E, H, dtype=torch.float64)
    msg_r = dict(kind=3, nbr_idx=idx, w_msg=f64(w_msg), b_msg=f64(b_msg), enc=f64(enc), src=f64(src), out=s_r, out2=hm_r,
                 next=dict(w=f64(mfc_w), b=f64(mfc_b), out=nxt_r))
    hr, cr = torch.empty(N, E, H, dtype=torch.float64), torch.empty(N, E, H, dtype=torch.float64)
    pir, actr = torch.zeros(N, E, A, dtype=torch.float64), torch.zeros(E, N, dtype=torch.uint8)
    gr = torch.zeros(N, E, 4 * H, dtype=torch.float64)
    ops_ref.lstm_step_policy(f64(h), f64(wh), f64(b), None, None, f64(c), f64(done), cr, hr, f64(pi_w), f64(pi_b), pir, actr,
                             xs=(None, f64(wx), None, None, msg_r), gates=gr, **draw)
    assert float(hm_r.abs().max()) > 0 and float((s_r - hm_r - enc.double()).abs().max()) < 1e-12
    img, mimg = ops.lstm_wimage(cu(wx), cu(wh)), ops.lstm_msg_wimage(cu(w_msg))
    save = torch.zeros(N, 3, E, H, device='cuda')                     # slots of a wider buffer: agent stride 3 E H
    msg_g = dict(kind=3, nbr_idx=cu(idx), w_msg=cu(w_msg), b_msg=cu(b_msg), img=mimg, enc=cu(enc), src=cu(src), out=save[:, 0],
                 out2=save[:, 2])
    nxt_g = torch.zeros(N, 2, E, H, device='cuda')                    # the sender layer on the new h: slot 1 of a wider buffer
    tol = dict(rtol=5e-5, atol=1e-5)
    for inplace in (False, True):
        save.zero_()
        nxt_g.zero_()
        if inplace:
            msg_g = dict(msg_g, next=dict(img=ops.lstm_msg_wimage(cu(mfc_w)), b=cu(mfc_b), out=nxt_g[:, 1]))
        hin, cin = cu(h), cu(c)
        hg, cg = (hin, cin) if inplace else (torch.zeros(N, E, H, device='cuda'), torch.zeros(N, E, H, device='cuda'))
        pig, actg, gg = torch.zeros(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda'), torch.zeros(N, E, 4 * H, device='cuda')
        ops.lstm_step_policy(hin, None, cu(b), None, None, cin, cu(done), cg, hg, cu(pi_w), cu(pi_b), pig, actg,
                             xs=(None, None, img, None, msg_g), gates=gg, **draw)
        torch.testing.assert_close(save[:, 2].cpu().double(), hm_r, **tol)
        torch.testing.assert_close(save[:, 0].cpu().double(), s_r, **tol)
        assert float(save[:, 1].abs().max()) == 0.0
        torch.testing.assert_close(hg.cpu().double(), hr, **tol)
        torch.testing.assert_close(cg.cpu().double(), cr, **tol)
        torch.testing.assert_close(gg.cpu().double(), gr, **tol)
        torch.testing.assert_close(pig.cpu().double(), pir, **tol)
        if inplace:
            torch.testing.assert_close(nxt_g[:, 1].cpu().double(), nxt_r, **tol)
            assert float(nxt_r.abs().max()) > 0
        assert float(nxt_g[:, 0].abs().max()) == 0.0 and (inplace or float(nxt_g.abs().max()) == 0.0)
    msg_g.pop('next')
    act_chk = torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.sample_actions(pig.cpu(), act_chk, **draw)
    assert torch.equal(actg.cpu(), act_chk)
    # value re-step: new message vectors (the caller's fc launch on the new h), nothing stored
    src2 = torch.relu(r(N, E, H))
    vr = torch.zeros(N, E, dtype=torch.float64)
    ops_ref.lstm_step_value(hr, f64(wh), f64(b), None, None, cr, f64(done), torch.empty_like(cr), torch.empty_like(hr), f64(v_w),
                            f64(v_b), act_chk, idx, A, vr, xs=(None, f64(wx), None, None, dict(msg_r, src=f64(src2), out=None, out2=None, next=None)))
    vg, h2, c2 = torch.zeros(N, E, device='cuda'), torch.zeros_like(hg), torch.zeros_like(cg)
    keep = save.clone()
    ops.lstm_step_value(hg, None, cu(b), None, None, cg, cu(done), c2, h2, cu(v_w), cu(v_b), actg, cu(idx), A, vg,
                        xs=(None, None, img, None, dict(msg_g, src=cu(src2), out=None, out2=None)))
    torch.testing.assert_close(vg.cpu().double(), vr, rtol=2e-4, atol=5e-5)
    assert torch.equal(save, keep)
    # the one-launch lock-step does not exist for this message kind
    with pytest.raises(Exception):
        ops.lstm_step_policy_value(cu(h), None, cu(b), None, None, cu(c), cu(done), cu(pi_w), cu(pi_b), pig, actg, cu(v_w), cu(v_b),
                                   cu(idx), A, vg, xs=(None, None, img, None, dict(msg_g, sync=ops.step_sync_words(N, E, 'cuda'))),
                                   h_out=h2, c_out=c2, gates=gg, **draw)


@pytest.mark.parametrize('N,E,topo', [(8, 4096, 'line'), (5, 130, 'line'), (9, 200, 'grid'), (6, 77, 'ragged')])
def test_dial_msg_adjoint(N, E, topo):
    """lstm_dial's message adjoint of one reverse step in one launch (nmarl_dial_msg_adjoint) vs the float64 restatement built
    from the autograd adjoint of the neighbour gather: both relu masks, 2- and 4-source agents, agents without sources, ragged
    -1 padded tables (asymmetric), slices of wider [N,T,E,64] buffers as operands."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    g = torch.Generator().manual_seed(N * 7 + E)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    if topo == 'ragged':
        m_max = 2
        idx = -torch.ones(N, m_max, dtype=torch.int32)
        idx[0, 0], idx[1, 0], idx[1, 1], idx[2, 0], idx[4, 0], idx[4, 1] = 3, 0, 3, 3, 1, 3       # agent 3: four sources, 5: none at all
    else:
        idx, _ = ops.neighbor_table(_topology(N, topo), 'cpu')
        m_max = idx.shape[1]
    T = 3
    DS, HM, MSG = r(N, T, E, H), torch.relu(r(N, T, E, H)), torch.relu(r(N, T, E, H))
    dhd = r(N, E, H)
    w_msg, mfc_w = r(N, H * m_max, H) * 0.2, r(N, H, H) * 0.2
    f64 = lambda t: t.double()                                                           # noqa: E731
    d1r, d2r, dhr = (torch.zeros(N, E, H, dtype=torch.float64) for _ in range(3))
    ops_ref.dial_msg_adjoint(f64(DS[:, 1]), f64(HM[:, 1]), f64(MSG[:, 1]), f64(dhd), f64(w_msg), f64(mfc_w), idx, None, None, d1r, d2r, dhr)
    rev = ops.reverse_neighbor_table(idx.cuda(), ops.COUPLED_NC)
    assert ops.dial_adjoint_supported(m_max, H, rev)
    imgs = ops.dial_adjoint_images(w_msg.cuda(), mfc_w.cuda())
    DSg, HMg, MSGg = DS.cuda(), HM.cuda(), MSG.cuda()
    D1, D2 = torch.zeros(N, T, E, H, device='cuda'), torch.zeros(N, T, E, H, device='cuda')
    dh = torch.zeros(N, E, H, device='cuda')
    parts = ops.dial_adjoint_bias_parts(N, E, 'cuda')
    for rep in range(2):                                               # the bias sums accumulate over calls (= reverse steps)
        out = ops.dial_msg_adjoint(DSg[:, 1], HMg[:, 1], MSGg[:, 1], dhd.cuda(), w_msg.cuda(), mfc_w.cuda(), idx.cuda(), imgs, rev,
                                   D1[:, 1], D2[:, 1], dh, bias_parts=parts)
    assert out is dh
    torch.testing.assert_close(parts[0].sum(1).cpu().double(), 2 * d1r.sum(1), rtol=1e-4, atol=1e-4 * E ** 0.5)
    torch.testing.assert_close(parts[1].sum(1).cpu().double(), 2 * d2r.sum(1), rtol=1e-4, atol=1e-4 * E ** 0.5)
    D1.zero_(), D2.zero_(), dh.zero_()
    ops.dial_msg_adjoint(DSg[:, 1], HMg[:, 1], MSGg[:, 1], dhd.cuda(), w_msg.cuda(), mfc_w.cuda(), idx.cuda(), imgs, rev, D1[:, 1], D2[:, 1], dh)
    assert torch.equal(D1[:, 1].cpu().double(), d1r)                   # a mask: exact
    torch.testing.assert_close(D2[:, 1].cpu().double(), d2r, rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(dh.cpu().double(), dhr, rtol=2e-5, atol=1e-5)
    assert float(D1[:, 0].abs().max()) == 0.0 and float(D1[:, 2].abs().max()) == 0.0 and float(D2[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize('N,E,A', [(8, 4096, 4), (3, 77, 5)])
def test_onehot_argmax_add(N, E, A):
    """lstm_dial's own-action term (agents/utils.py:577-579) in one launch vs one_hot(argmax): first maximum on ties, per-agent
    scale (lstm_dial_hetero), a column block of a wider buffer as target."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(E)
    p = torch.softmax(torch.randn(N, E, A, generator=g), dim=-1)
    p[:, ::3] = 1.0 / A                                              # ties: index 0
    p[:, 1::7, A - 1] = p[:, 1::7, 1] = 0.45                         # ties between 1 and A-1: index 1
    y0 = torch.randn(N, E, 64, generator=g)
    scale = torch.tensor([float(i % 2) for i in range(N)])
    for sc in (None, scale):
        ref = ops_ref.onehot_argmax_add_(y0.clone(), p, sc)
        wide = torch.zeros(N, E, 192, device='cuda')
        wide[:, :, 64:128].copy_(y0)
        ops.onehot_argmax_add_(wide[:, :, 64:128], p.cuda(), None if sc is None else sc.cuda())
        assert torch.equal(wide[:, :, 64:128].cpu(), ref)
        assert float(wide[:, :, :64].abs().max()) == 0.0 and float(wide[:, :, 128:].abs().max()) == 0.0


def _topology(N, kind):
    """line (m_max 2) or 5x5-style grid (m_max 4) neighbour masks for N agents."""
    nm = np.zeros((N, N), dtype=int)
    if kind == 'line':
        for i in range(N - 1):
            nm[i, i + 1] = nm[i + 1, i] = 1
    else:
        side = int(round(N ** 0.5))
        assert side * side == N
        for i in range(N):
            r, c_ = divmod(i, side)
            for rr, cc in ((r - 1, c_), (r + 1, c_), (r, c_ - 1), (r, c_ + 1)):
                if 0 <= rr < side and 0 <= cc < side:
                    nm[i, rr * side + cc] = 1
    return nm


@pytest.mark.parametrize('kind,topo,N,T,E', [(1, 'line', 8, 12, 4096), (1, 'line', 8, 60, 300), (1, 'line', 3, 5, 127),
                                             (2, 'line', 8, 12, 4096), (2, 'grid', 25, 7, 1024), (2, 'grid', 9, 6, 130),
                                             (1, 'line', 2, 3, 1)])
def test_lstm_bptt_coupled_one_launch(kind, topo, N, T, E):
    """nmarl_lstm_bptt_coupled: the whole reverse recurrence of a coupled net (lstm_comm kind 1 / lstm_ic3 kind 2,
    agents/utils.py:182-208, 395-408) in one launch, the message adjoint handed between the agents' blocks inside the kernel
    -- against the float64 restatement (oracle/ops_ref.py bptt_coupled: cell backward, dgrad, relu mask, neighbour gather /
    mean adjoint per step), and the step-wise form of the same kernel (mode 2: T launches) bit for bit.  The one-launch
    form is forced (mode 1) where the grid fits the chip, so the in-kernel hand-off (flags, write-through ring) is what runs;
    strided sequence buffers, ragged rows, dones inside the sequence."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H = 64
    nm = _topology(N, topo)
    nbr_idx, _ = ops.neighbor_table(nm, 'cuda')
    m_max = nbr_idx.shape[1]
    K = H * m_max if kind == 1 else H
    g = torch.Generator().manual_seed(N * 131 + T * 7 + E + kind)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    gates = torch.cat([torch.sigmoid(r(N, T, E, 3 * H)), torch.tanh(r(N, T, E, H))], dim=-1)
    c0 = r(N, E, H) * 0.8
    done = (torch.rand(T, E, generator=g) < 0.2).float()
    call = _forward_cells(gates, c0, done)
    dhs = r(N, T, E, H)
    wh = r(N, H, 4 * H) * 0.1
    wxm = r(N, H, 4 * H) * 0.1
    w_msg = r(N, K, H) * 0.15
    S = torch.relu(r(N, T, E, 3 * H))                     # lstm_comm's saved LSTM input; its last third is the message term hm
    hm = S[..., 2 * H:]
    dz_r = torch.empty(N, T, E, 4 * H, dtype=torch.float64)
    d1_r = torch.empty(N, T, E, H, dtype=torch.float64)
    rev_r = ops_ref.reverse_neighbor_table(nbr_idx.cpu(), kind)
    db_r, dbm_r = ops_ref.bptt_coupled(kind, rev_r, m_max, gates.double(), call.double(), done.double(), dhs.double(),
                                       (wxm.double(), wh.double(), None), (w_msg.double(), None),
                                       hm.double() if kind == 1 else None, dz_r, d1_r)
    # device buffers: slots of wider allocations
    G = torch.zeros(N, T + 2, E, 4 * H, device='cuda'); G[:, 1:T + 1].copy_(gates)
    C = torch.zeros(N, T + 3, E, H, device='cuda'); C[:, 1:T + 2].copy_(call)
    D = torch.zeros(N, T + 1, E, H, device='cuda'); D[:, :T].copy_(dhs)
    Sg = S.cuda()
    doneg = done.cuda()
    wxg, whg, wmg = wxm.cuda(), wh.cuda(), w_msg.cuda()
    ws = (wxg, whg, ops.lstm_bptt_wimage(wxg, whg))
    wm = (wmg, ops.lstm_bptt_msg_wimage(wmg))
    rev = ops.reverse_neighbor_table(nbr_idx, kind)
    assert rev is not None and rev['symmetric']
    out = {}
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    resident = N * -(-E // 128) <= cus
    for mode in ((1, 2) if resident else (0, 2)):
        dZ = torch.zeros(N, T + 2, E, 4 * H, device='cuda')
        D1 = torch.zeros(N, T, E, H, device='cuda')
        db, dbm = ops.bptt_coupled(kind, rev, m_max, G[:, 1:T + 1], C[:, 1:T + 2], doneg, D[:, :T], ws, wm,
                                   Sg[..., 2 * H:] if kind == 1 else None, dZ[:, 1:T + 1], D1, mode=mode)
        torch.cuda.synchronize()
        ops.check_coupled_status()
        assert torch.all(dZ[:, 0] == 0) and torch.all(dZ[:, T + 1] == 0)
        out[mode] = (dZ[:, 1:T + 1].clone(), D1, db, dbm)
    a, b = out[1 if resident else 0], out[2]
    for name, o in (('one launch', a), ('step-wise', b)):          # each form against the restatement first: says WHICH is off
        bad = (~torch.isclose(o[0].cpu().double(), dz_r, rtol=2e-4, atol=5e-5)).sum().item()
        assert bad == 0, '%s: %d of %d dz entries off the restatement' % (name, bad, dz_r.numel())
    ndiff = (a[0] != b[0]).sum().item() + (a[1] != b[1]).sum().item()
    assert ndiff == 0, 'one launch and step-wise launches differ in %d entries' % ndiff
    torch.testing.assert_close(a[2], b[2], rtol=1e-5, atol=1e-5 * (T * E) ** 0.5)
    dZ, D1, db, dbm = a
    torch.testing.assert_close(dZ.cpu().double(), dz_r, rtol=2e-4, atol=5e-5)
    torch.testing.assert_close(D1.cpu().double(), d1_r, rtol=2e-4, atol=5e-5)
    torch.testing.assert_close(db.cpu().double(), db_r, rtol=1e-4, atol=1e-4 * max(1.0, float(db_r.abs().max())))
    torch.testing.assert_close(dbm.cpu().double(), dbm_r, rtol=1e-4, atol=1e-4 * max(1.0, float(dbm_r.abs().max())))
    torch.testing.assert_close(db.cpu().double(), dZ.double().sum(dim=(1, 2)).cpu(), rtol=1e-5,
                               atol=1e-6 * max(1.0, float(db_r.abs().max())) * (T * E) ** 0.5)


@pytest.mark.parametrize('kind,topo,N,T,E,O', [(1, 'line', 8, 12, 4096, 5), (1, 'line', 3, 5, 127, 5), (2, 'grid', 25, 6, 1024, 6),
                                               (2, 'line', 8, 7, 300, 5), (2, 'grid', 9, 4, 77, 8)])
def test_lstm_bptt_coupled_expands_the_heads_gradient_itself(kind, topo, N, T, E, O):
    """nmarl_lstm_bptt_coupled with the heads' dL/dh handed over as dy8 + the heads' weights (round 6; formed inside the launch by two
    more k-steps of every step's transposed product, the operands in registers) against the same call on the tensor dL/dh = dy hw^T:
    dz, D1 and both bias gradients at fp32 summation-order tolerance, in the one-launch form and in step-wise launches (whose
    carried state crosses launches), which must agree with each other bit for bit."""
    from deeprl_network_amd import ops
    H = 64
    nbr_idx, _ = ops.neighbor_table(_topology(N, topo), 'cuda')
    m_max = nbr_idx.shape[1]
    K = H * m_max if kind == 1 else H
    g = torch.Generator().manual_seed(N * 31 + T * 5 + E + kind + O)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    gates = torch.cat([torch.sigmoid(r(N, T, E, 3 * H)), torch.tanh(r(N, T, E, H))], dim=-1)
    done = (torch.rand(T, E, generator=g) < 0.2).float()
    call = _forward_cells(gates, r(N, E, H) * 0.8, done)
    dy8 = torch.zeros(N, T * E, 8)
    dy8[:, :, :O] = r(N, T * E, O)
    hw = r(N, H, O) * 0.3
    dhs = torch.bmm(dy8[:, :, :O].double(), hw.double().transpose(1, 2)).float().view(N, T, E, H)
    wxg, whg, wmg = (r(N, H, 4 * H) * 0.1).cuda(), (r(N, H, 4 * H) * 0.1).cuda(), (r(N, K, H) * 0.15).cuda()
    Sg = torch.relu(r(N, T, E, 3 * H)).cuda()
    G, C, doneg = gates.cuda(), call.cuda(), done.cuda()
    ws = (wxg, whg, ops.lstm_bptt_wimage(wxg, whg))
    wm = (wmg, ops.lstm_bptt_msg_wimage(wmg))
    rev = ops.reverse_neighbor_table(nbr_idx, kind)
    resident = N * -(-E // 128) <= torch.cuda.get_device_properties(0).multi_processor_count
    out = {}
    for form, head_dy in (('tensor', None), ('dy8', (dy8.cuda(), hw.cuda()))):
        for mode in ((1, 2) if resident else (0, 2)):
            dZ, D1 = torch.zeros(N, T, E, 4 * H, device='cuda'), torch.zeros(N, T, E, H, device='cuda')
            db, dbm = ops.bptt_coupled(kind, rev, m_max, G, C, doneg, dhs.cuda() if head_dy is None else None, ws, wm,
                                       Sg[..., 2 * H:] if kind == 1 else None, dZ, D1, mode=mode, head_dy=head_dy)
            torch.cuda.synchronize()
            ops.check_coupled_status()
            out[form, mode] = (dZ, D1, db, dbm)
    m1 = 1 if resident else 0
    for x, y in zip(out['dy8', m1][:2], out['dy8', 2][:2]):
        assert torch.equal(x, y), 'dy8 form: one launch and step-wise launches differ'
    for x, y in zip(out['dy8', m1], out['tensor', m1]):
        torch.testing.assert_close(x, y, rtol=5e-5, atol=5e-6 * float(y.abs().max()))


def test_lstm_bptt_coupled_repeated_calls_under_load():
    """The in-kernel hand-off must not depend on timing: the same update twice while another stream keeps the chip busy
    with a bandwidth-bound kernel (uneven load between the agents' blocks), results identical to the quiet run."""
    from deeprl_network_amd import ops
    H, N, T, E, kind = 64, 8, 20, 4096, 1
    nbr_idx, _ = ops.neighbor_table(_topology(N, 'line'), 'cuda')
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).cuda()                                  # noqa: E731
    G = torch.cat([torch.sigmoid(r(N, T, E, 3 * H)), torch.tanh(r(N, T, E, H))], dim=-1)
    C, D = r(N, T + 1, E, H) * 0.8, r(N, T, E, H)
    done = (torch.rand(T, E, generator=g) < 0.1).float().cuda()
    wx, wh, wmsg = r(N, H, 4 * H) * 0.1, r(N, H, 4 * H) * 0.1, r(N, 2 * H, H) * 0.15
    S = torch.relu(r(N, T, E, 3 * H))
    ws, wm = (wx, wh, ops.lstm_bptt_wimage(wx, wh)), (wmsg, ops.lstm_bptt_msg_wimage(wmsg))
    rev = ops.reverse_neighbor_table(nbr_idx, kind)

    def run():
        dZ, D1 = torch.empty(N, T, E, 4 * H, device='cuda'), torch.empty(N, T, E, H, device='cuda')
        db, dbm = ops.bptt_coupled(kind, rev, 2, G, C, done, D, ws, wm, S[..., 2 * H:], dZ, D1, mode=1)
        return dZ, D1, db.clone(), dbm.clone()
    quiet = run()
    torch.cuda.synchronize()
    ops.check_coupled_status()
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device='cuda')
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(20):
                big.mul_(1.0001)
        loaded = run()
        torch.cuda.synchronize()
        ops.check_coupled_status()
        for x, y in zip(quiet, loaded):
            assert torch.equal(x, y)


@pytest.mark.parametrize('N,E,H,A,F,T', [(8, 4096, 64, 4, 5, 60), (25, 130, 64, 5, 12, 7), (3, 1, 64, 4, 15, 3)])
def test_batch_epilogue_matches_host_code(N, E, H, A, F, T):
    """nmarl_batch_epilogue (episode statistics + state hand-over between two batches, two launches) == the elementwise
    host code it replaces (oracle/ops_ref.py batch_epilogue), over two consecutive batches with episodes ending in the
    first (some early = collisions)."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    g_ = torch.Generator().manual_seed(N + E)
    T_env = 3 * T
    host = dict(ep_sum=torch.zeros(E, dtype=torch.float64), ep_sq=torch.zeros(E, dtype=torch.float64),
                ep_len=torch.full((E,), float(T), dtype=torch.float64), fin=torch.zeros(4, dtype=torch.float64),
                h_fw=torch.randn(N, E, H, generator=g_), c_fw=torch.randn(N, E, H, generator=g_),
                h_bw=torch.zeros(N, E, H), c_bw=torch.zeros(N, E, H), fp_0=torch.zeros(N, E, A), x_0=torch.zeros(E, N, F),
                done_pre=torch.zeros(E))
    fp_uniform = torch.rand(N, 1, A, generator=g_)
    dev = {k: v.clone().cuda() for k, v in host.items()}
    for b in range(2):
        g = -torch.rand(T, E, generator=g_) * 100
        done = (torch.rand(E, generator=g_) < (0.4 if b == 0 else 1.0)).to(torch.uint8)
        fp_T, x_T = torch.rand(N, E, A, generator=g_), torch.randn(E, N, F, generator=g_)
        ops_ref.batch_epilogue(g, done, host['ep_sum'], host['ep_sq'], host['ep_len'], host['fin'], T_env, host['h_fw'],
                               host['c_fw'], host['h_bw'], host['c_bw'], fp_T, host['fp_0'], fp_uniform, x_T, host['x_0'],
                               host['done_pre'])
        ops.batch_epilogue(g.cuda(), done.cuda(), dev['ep_sum'], dev['ep_sq'], dev['ep_len'], dev['fin'], T_env, dev['h_fw'],
                           dev['c_fw'], dev['h_bw'], dev['c_bw'], fp_T.cuda(), dev['fp_0'], fp_uniform.cuda(), x_T.cuda(),
                           dev['x_0'], dev['done_pre'])
        for k in ('h_fw', 'c_fw', 'h_bw', 'c_bw', 'fp_0', 'x_0', 'done_pre'):
            assert torch.equal(dev[k].cpu(), host[k]), k
        for k in ('ep_sum', 'ep_sq', 'ep_len', 'fin'):
            torch.testing.assert_close(dev[k].cpu(), host[k], rtol=1e-12, atol=1e-9)
    assert host['fin'][0] > 0 and (E == 1 or host['fin'][3] > 0)


@pytest.mark.parametrize('N,E', [(8, 4096), (8, 1000), (8, 77), (5, 1), (3, 129)])
@pytest.mark.parametrize('mode', [1, 2])
def test_lstm_step_x_input_encoders_inside_the_launch(N, E, mode):
    """nmarl_lstm_step_x_enc (lstm_step_x_kernel<3,0,1>): FPPolicy's two input encoders (policies.py:176-181) as the
    register-only pre-phase of the policy + value launch, from the compact observation [E,N,5] and the previous-step policies
    [N,E,4] through the neighbour table (ascending, left packed, absent slots zero).  Against the float64 restatement
    (encoder output, new state, gates, policy, draw, value) and against the separate encoder launch (nmarl_fc_fwd_multi)
    followed by the same step on its output; ragged row counts, agents with 0 / 1 / 2 neighbours, strided output slot."""
    from deeprl_network_amd import ops
    from oracle import ops_ref
    H, A, KX = 64, 4, 128
    g = torch.Generator().manual_seed(N * 131 + E + mode)
    r = lambda *s: torch.randn(*s, generator=g)                                         # noqa: E731
    d = _xside_case(N, E, KX, A, 2, N * 7 + E, 0)
    # line graph (agents at the ends have one neighbour), one isolated agent when N = 3
    nbrs = [[j for j in (i - 1, i + 1) if 0 <= j < N] for i in range(N)]
    if N == 3:
        nbrs = [[1], [0], []]
    ob, fp = r(E, N, 5), torch.softmax(r(N, E, A), dim=-1)
    w_ob, b_ob = r(N, 15, H) * 0.4 + torch.arange(15).view(1, -1, 1) * 0.01, r(N, H) * 0.2
    w_fp, b_fp = r(N, 8, H) * 0.4 + torch.arange(H).view(1, 1, -1) * 0.003, r(N, H) * 0.2
    for i in range(N):                      # rows of absent slots are zero in the product's padded weights
        w_ob[i, 5 * (1 + len(nbrs[i])):] = 0
        w_fp[i, 4 * len(nbrs[i]):] = 0
    f64 = lambda t: None if t is None else t.double()                                    # noqa: E731
    cu = lambda t: None if t is None else t.cuda()                                       # noqa: E731
    draw = dict(mode=mode, seed=9, env_id_base=17, step=5)
    # float64 restatement
    spec_r = ops_ref.step_enc_spec(f64(ob), f64(fp), f64(w_ob), f64(b_ob), f64(w_fp), f64(b_fp), nbrs)
    S_r = ops_ref.step_enc_forward(spec_r)
    hr, cr = d['h'].double(), d['c'].double()
    gr = torch.empty(N, E, 4 * H, dtype=torch.float64)
    ops_ref.lstm_step_fused(hr, f64(d['wh']), f64(d['b']), None, None, cr, f64(d['done']), gr, torch.empty_like(cr),
                            torch.empty_like(hr), xs=(S_r, f64(d['wx']), None))
    pir, actr = torch.zeros(N, E, A, dtype=torch.float64), torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.lstm_step_policy(hr, f64(d['wh']), f64(d['b']), None, None, cr, f64(d['done']), cr, hr, f64(d['pi_w']), f64(d['pi_b']),
                             pir, actr, xs=(S_r, f64(d['wx']), None), **draw)
    # the kernel: output into a strided slot of a wider sequence buffer
    img = ops.lstm_wimage(cu(d['wx']), cu(d['wh']))
    Sbuf = torch.full((N, 3, E, KX), -7.0, device='cuda')
    ob_g, fp_g = cu(ob), cu(fp)
    bits = torch.full((N, 3, E, 4), 0x5a5a5a5a, dtype=torch.int32, device='cuda')
    spec = ops.step_enc_spec(ob_g, fp_g, cu(w_ob), cu(b_ob), cu(w_fp), cu(b_fp), nbrs, out=Sbuf[:, 1], bits=bits[:, 1])
    hg, cg = cu(d['h']), cu(d['c'])
    ho, co, gg = torch.empty_like(hg), torch.empty_like(cg), torch.empty(N, E, 4 * H, device='cuda')
    pig, actg, vg = torch.zeros(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda'), torch.zeros(N, E, device='cuda')
    ops.lstm_step_policy_value(hg, None, cu(d['b']), None, None, cg, cu(d['done']), cu(d['pi_w']), cu(d['pi_b']), pig, actg,
                               cu(d['v_w']), cu(d['v_b']), cu(d['idx']), A, vg, xs=(spec, None, img), h_out=ho, c_out=co, gates=gg,
                               defer_action_term=True, **draw)
    torch.cuda.synchronize()
    assert torch.all(Sbuf[:, 0] == -7.0) and torch.all(Sbuf[:, 2] == -7.0)
    # the sign image of the encoder outputs (what the update's encoder backward reads instead of S): exactly (S > 0), packed
    assert torch.equal(bits[:, 1], ops.relu_bits_pack(Sbuf[:, 1])) and torch.all(bits[:, 0] == 0x5a5a5a5a) and torch.all(bits[:, 2] == 0x5a5a5a5a)
    assert torch.equal(bits[:, 1].cpu(), ops_ref.relu_bits_pack(Sbuf[:, 1].cpu()))
    tol = dict(rtol=3e-5, atol=5e-6)
    torch.testing.assert_close(Sbuf[:, 1].cpu().double(), S_r, **tol)
    torch.testing.assert_close(ho.cpu().double(), hr, **tol)
    torch.testing.assert_close(co.cpu().double(), cr, **tol)
    torch.testing.assert_close(gg.cpu().double(), gr, **tol)
    torch.testing.assert_close(pig.cpu().double(), pir, **tol)
    act_chk = torch.zeros(E, N, dtype=torch.uint8)
    ops_ref.sample_actions(pig.cpu(), act_chk, **draw)
    assert torch.equal(actg.cpu(), act_chk)
    # against the separate encoder launch + the same step on its output (incl. the value re-step, whose h part is deferred)
    nbr_idx = -torch.ones(N, 2, dtype=torch.int32)
    for i, lst in enumerate(nbrs):
        nbr_idx[i, :len(lst)] = torch.tensor(lst, dtype=torch.int32)
    nbr_self = torch.cat([torch.arange(N, dtype=torch.int32).view(-1, 1), nbr_idx], dim=1).cuda()
    S2 = ops.fc_fwd_multi([(ob_g.transpose(0, 1), cu(w_ob), cu(b_ob), nbr_self), (fp_g, cu(w_fp), cu(b_fp), nbr_idx.cuda())], ops.BIAS_RELU)
    torch.testing.assert_close(Sbuf[:, 1], S2, rtol=1e-5, atol=1e-6)
    h2, c2, g2 = torch.empty_like(hg), torch.empty_like(cg), torch.empty_like(gg)
    pi2, act2, v2 = torch.zeros_like(pig), torch.zeros_like(actg), torch.zeros_like(vg)
    ops.lstm_step_policy_value(hg, None, cu(d['b']), None, None, cg, cu(d['done']), cu(d['pi_w']), cu(d['pi_b']), pi2, act2,
                               cu(d['v_w']), cu(d['v_b']), cu(d['idx']), A, v2, xs=(S2, None, img), h_out=h2, c_out=c2, gates=g2,
                               defer_action_term=True, **draw)
    torch.testing.assert_close(ho, h2, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(vg, v2, rtol=1e-4, atol=2e-5)
    assert (actg != act2).float().mean().item() < 1e-3                 # a draw flips only where a uniform meets a CDF boundary
    # no output slot (bootstrap step): same results, nothing written
    pi3, act3, v3 = torch.zeros_like(pig), torch.zeros_like(actg), torch.zeros_like(vg)
    h3, c3 = torch.empty_like(hg), torch.empty_like(cg)
    spec3 = ops.step_enc_spec(ob_g, fp_g, cu(w_ob), cu(b_ob), cu(w_fp), cu(b_fp), nbrs, out=None)
    ops.lstm_step_policy_value(hg, None, cu(d['b']), None, None, cg, cu(d['done']), cu(d['pi_w']), cu(d['pi_b']), pi3, act3,
                               cu(d['v_w']), cu(d['v_b']), cu(d['idx']), A, v3, xs=(spec3, None, img), h_out=h3, c_out=c3,
                               defer_action_term=True, **draw)
    assert torch.equal(h3, ho) and torch.equal(v3, vg) and torch.equal(act3, actg)


@pytest.mark.parametrize('N,rows,gather', [(8, 60 * 4096, True), (3, 1000, True), (2, 64 * 33 + 5, False), (5, 7, False)])
def test_fc_bwd_pair_equals_two_fc_bwd_launches(N, rows, gather):
    """nmarl_fc_bwd_pair: both layers of [relu(x_0 w_0 + b_0) | relu(x_1 w_1 + b_1)] in ONE pass over dS -- bit-identical to
    two nmarl_fc_bwd(_gather) launches (same partition of the rows, same order of every sum), with the relu derivative taken
    from S or from the 16-byte-per-row sign image (relu_bits_pack's layout, what the lock-step kernel's encoders write);
    BASELINE size, ragged row counts, gathered and plain inputs, tanh."""
    from deeprl_network_amd import ops
    g = torch.Generator().manual_seed(N * 977 + rows)
    r = lambda *s: torch.randn(*s, generator=g).cuda()                                    # noqa: E731
    if gather:
        nbrs = [[j for j in (i - 1, i + 1) if 0 <= j < N] for i in range(N)]
        nbr_idx = -torch.ones(N, 2, dtype=torch.int32)
        for i, lst in enumerate(nbrs):
            nbr_idx[i, :len(lst)] = torch.tensor(lst, dtype=torch.int32)
        nbr_self = torch.cat([torch.arange(N, dtype=torch.int32).view(-1, 1), nbr_idx], dim=1).cuda()
        xs = [r(rows, N, 5).transpose(0, 1), torch.softmax(r(N, rows, 4), dim=-1)]      # the compact slab [rows,N,5] read in place
        idxs = [nbr_self, nbr_idx.cuda()]
        Fs = [15, 8]
    else:
        xs, idxs, Fs = [r(N, rows, 16), r(N, rows, 3)], [None, None], [16, 3]
    ws = [r(N, F, 64) * 0.3 for F in Fs]
    bs = [r(N, 64) * 0.1 for _ in Fs]
    for act in (ops.BIAS_RELU, ops.BIAS_TANH):
        S = ops.fc_fwd_multi([(x, w, b, i) for x, w, b, i in zip(xs, ws, bs, idxs)], act)
        dS = r(N, rows, 128)
        two = [ops.fc_bwd(x, S[:, :, 64 * k:64 * k + 64], dS[:, :, 64 * k:64 * k + 64], act, nbr_idx=i) for k, (x, i) in enumerate(zip(xs, idxs))]
        assert ops.fc_bwd_pair_supported(xs, idxs, S, dS)
        variants = [ops.fc_bwd_pair(xs, idxs, S, dS, act)]
        if act == ops.BIAS_RELU:
            variants.append(ops.fc_bwd_pair(xs, idxs, None, dS, act, bits=ops.relu_bits_pack(S)))
        for one in variants:
            for (dw2, db2), (dw1, db1) in zip(two, one):
                assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
        # and against the definition (float64)
        if not gather:
            xg = xs
            d = (S > 0).double() if act == ops.BIAS_RELU else 1.0 - S.double() ** 2
            G = dS.double() * d
            for k in range(2):
                torch.testing.assert_close(variants[0][k][0].double(), torch.bmm(xg[k].double().transpose(1, 2), G[:, :, 64 * k:64 * k + 64]),
                                           rtol=1e-4, atol=1e-3 * max(1.0, rows ** 0.5 / 30))
