"""Batched trainer on the GPU at the BASELINE size (8 agents x 4096 replicas): size-independent properties of the
whole rollout + update path -- hipGraph replay == eager launch, run-to-run determinism (no atomics anywhere),
batch invariance of the rollout (replica e of 4096 == the same replica in a batch of 16), finite updates."""
import numpy as np
import pytest
import torch

from helpers import cacc_config

pytestmark = pytest.mark.gpu


def build(agent, E, use_graph, env_id_base=0, scenario='catchup', n_step=60, episode_sec=None, **trainer_kw):
    """scenario 'grid': BASELINE configs[3]'s environment (5 x 5 synthetic ATSC grid, config_ma2c_cnet_grid.ini), n_step as given."""
    from deeprl_network_amd.envs import make_batch_env
    from deeprl_network_amd.main import init_agent
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    if scenario == 'grid':
        from helpers import grid_config
        cp = grid_config(agent=agent, n_step=n_step)
    else:
        cp = cacc_config(agent=agent, scenario=scenario, n_step=n_step, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
    if episode_sec is not None:
        cp['ENV_CONFIG']['episode_length_sec'] = str(episode_sec)
    env = make_batch_env(cp['ENV_CONFIG'], num_envs=E, env_id_base=env_id_base)
    np.random.seed(12)
    model = init_agent(env, cp['MODEL_CONFIG'], 10 ** 9, 12, num_envs=E)
    return env, model, BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=use_graph, **trainer_kw)


# (agent, scenario, replicas, n_step): BASELINE configs[1], [2] and [3]
BASELINE_CASES = [('ia2c_fp', 'catchup', 4096, 60), ('ma2c_nc', 'slowdown', 4096, 60), ('ma2c_ic3', 'grid', 1024, 120)]


@pytest.mark.parametrize('agent,scenario,E,n_step', BASELINE_CASES)
def test_graph_equals_eager_and_is_deterministic(agent, scenario, E, n_step):
    runs = []
    for use_graph in (True, False, True):
        env, model, tr = build(agent, E, use_graph, scenario=scenario, n_step=n_step)
        for _ in range(3):
            tr.run_batch()
        torch.cuda.synchronize()
        assert tr.handoff_fallbacks == 0
        runs.append((model.policy.params.flat.clone(), env.state_tensors()[0].clone(), model.buf_act.clone(), tr.R_end.clone()))
        del env, model, tr
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b), 'hipGraph replay differs from eager launches'
    for a, b in zip(runs[0], runs[2]):
        assert torch.equal(a, b), 'two identical runs differ (non-deterministic kernel?)'
    assert torch.isfinite(runs[0][0]).all()


@pytest.mark.parametrize('agent,scenario', [('ia2c_fp', 'catchup'), ('ma2c_nc', 'catchup'), ('ma2c_ic3', 'catchup'), ('ma2c_cu', 'catchup'),
                                            ('ma2c_dial', 'catchup'), ('ma2c_ic3', 'grid')])
def test_captured_update_equals_eager_update(agent, scenario, monkeypatch):
    """The A2C update replayed as a hipGraph (from the second batch on; rewards, return scan, loss, backward, clip + RMSProp,
    and for nets without the hand-off guard the batch epilogue) against the eager update behind the same rollout graph: weights,
    RMSProp slots, env state, actions, episode statistics bit-identical after 4 batches -- with a LINEAR lr schedule, whose
    value reaches the captured optimiser step through a device scalar.  ('grid': BASELINE configs[3], 25 x 1024, whose update
    runs lstm_bptt_coupled_kernel<4,4,false>.)"""
    E = 1024
    runs = []
    for capture in ('1', '0'):
        monkeypatch.setenv('NMARL_CAPTURE_UPDATE', capture)
        env, model, tr = build(agent, E, True, n_step=20, scenario=scenario)
        model.lr_scheduler = type(model.lr_scheduler)(5e-4, 1e-5, 200, decay='linear')
        for _ in range(4):
            tr.run_batch()
        torch.cuda.synchronize()
        assert (tr._upd is not None) == (capture == '1'), tr.update_capture_error
        assert tr.update_capture_error is None and tr.handoff_fallbacks == 0
        runs.append((model.policy.params.flat.clone(), model.policy.params.ms.clone(), env.state_tensors()[0].clone(), model.buf_act.clone(),
                     tr.R_end.clone(), tr.ep_sum.clone(), model.h_bw.clone(), torch.tensor(model.cur_lr)))
        del env, model, tr
    for k, (a, b) in enumerate(zip(*runs)):
        assert torch.equal(a, b), 'captured update differs from the eager one (item %d)' % k
    assert float(runs[0][7]) == pytest.approx(5e-4 * (1 - 80 / 200))


@pytest.mark.parametrize('use_graph', [True, False])
def test_commnet_update_uses_the_saved_encoder_outputs_every_batch(use_graph, monkeypatch):
    """CommNet: the rollout saves the encoder outputs of every lock-step, and the update must use them in EVERY batch -- also
    when the rollout is a hipGraph replay (no Python runs then: the per-batch `_enc_was_saved` flag the update clears has to be
    raised again by the trainer).  The recomputing encoder forward is counted; same weights as the eager run either way."""
    from deeprl_network_amd.agents.policies import IC3MultiAgentPolicy
    calls = {'n': 0}
    orig = IC3MultiAgentPolicy._enc

    def counting(self, xv, fp):
        calls['n'] += 1
        return orig(self, xv, fp)
    monkeypatch.setattr(IC3MultiAgentPolicy, '_enc', counting)
    from deeprl_network_amd import ops
    means = {'n': 0}
    orig_mean = ops.nbr_mean

    def counting_mean(x, nbr_idx):
        means['n'] += x.shape[1] > 1024                     # an averaging pass over the whole h sequence (the update's)
        return orig_mean(x, nbr_idx)
    monkeypatch.setattr(ops, 'nbr_mean', counting_mean)
    env, model, tr = build('ma2c_ic3', 1024, use_graph)
    for _ in range(4):
        tr.run_batch()
    torch.cuda.synchronize()
    assert model.save_acts and 'ENC' in model.policy._extra and 'MM' in model.policy._extra
    assert calls['n'] == 0, 'the update recomputed the encoder forward %d times' % calls['n']
    assert means['n'] == 0, 'the update averaged the h sequence %d times although the rollout kept the means' % means['n']
    assert float(model.policy._extra_full['MM'][:, -1].abs().max()) == 0.0          # the padding slab stays zero
    assert torch.isfinite(model.policy.params.flat).all()


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_ic3'])
def test_rollout_batch_invariance_at_full_size(agent):
    env, model, tr = build(agent, 4096, False)
    tr._rollout()
    senv, smodel, str_ = build(agent, 16, False, env_id_base=2000)
    str_._rollout()
    sl = slice(2000, 2016)
    assert torch.equal(model.buf_act[:, sl], smodel.buf_act), 'sampled actions depend on the batch size'
    torch.testing.assert_close(model.buf_v[:, :, sl], smodel.buf_v, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(tr.R_end[:, sl], str_.R_end, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(tr.buf_rraw[:, sl], str_.buf_rraw, rtol=1e-5, atol=1e-3)
    assert torch.equal(model.buf_done_post[:, sl], smodel.buf_done_post)


def test_episode_bookkeeping_on_gpu():
    """T = 3 batches: after 3 batches every replica finished one episode and restarted clean."""
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    cp = cacc_config(agent='ma2c_cu', n_step=10, reward_norm=5000.0)
    cp['ENV_CONFIG']['episode_length_sec'] = '3'
    env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=512)
    np.random.seed(3)
    model = models.IA2C_CU(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                           cp['MODEL_CONFIG'], seed=3, num_envs=512)
    tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
    for _ in range(3):
        tr.run_batch()
    st = tr.stats()
    assert st['episodes'] == 512 and torch.all(env.episode == 2) and torch.all(tr.done_pre == 1)
    assert torch.all(model.h_fw == 0) and torch.allclose(model.fp, torch.full_like(model.fp, 0.25))
    m, s, c = tr.evaluate(n_envs=32)
    assert np.isfinite(m) and 0 <= c <= 32


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc'])
def test_one_pass_encoder_backward_equals_two_launches(agent, monkeypatch):
    """The update's encoder backward as ONE pass over dS (nmarl_fc_bwd_pair; IA2C-FP: the relu derivative from the 16-byte sign
    image the lock-step kernel wrote, S itself not read) against the two fc_bwd launches: weights, optimiser slots and actions
    bit-identical after 4 batches at the BASELINE size (hipGraph rollout and update).  (NeurComm's one-launch lock-step writes the
    sign image as well, round 6.)"""
    runs = []
    for pair in ('1', '0'):
        monkeypatch.setenv('NMARL_FC_BWD_PAIR', pair)
        env, model, tr = build(agent, 4096, True)
        for _ in range(4):
            tr.run_batch()
        torch.cuda.synchronize()
        assert (model.S_bits is not None) == (pair == '1') and tr.enc_in_kernel
        runs.append((model.policy.params.flat.clone(), model.policy.params.ms.clone(), model.buf_act.clone(), tr.R_end.clone()))
        del env, model, tr
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_graphs_equal_eager_over_many_batches_with_evaluations_in_between():
    """12 batches: rollout + update graphs (with a test episode before the first batch and after every third: its graph shares
    the device with the trainer's) against eager launches without any evaluation -- weights, optimiser slots, env state and
    actions bit-identical.  Evaluation touches nothing the training reads."""
    runs = []
    for use_graph in (True, False):
        env, model, tr = build('ia2c_fp', 4096, use_graph)
        if use_graph:
            tr.evaluate(n_envs=64)
        for b in range(12):
            tr.run_batch()
            if use_graph and b % 3 == 2:
                tr.evaluate(n_envs=64)
        torch.cuda.synchronize()
        runs.append((model.policy.params.flat.clone(), model.policy.params.ms.clone() if hasattr(model.policy.params, 'ms') else env.v.clone(),
                     env.h.clone(), model.buf_act.clone(), tr.R_end.clone()))
        del env, model, tr
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc'])
def test_evaluation_graph_built_before_training_stays_right(agent):
    """The test-episode hipGraph captured BEFORE the first training batch, replayed after the trainer's rollout and update graphs
    exist and the weights moved: reward sums, episode lengths and the action histogram equal the same episode launched eagerly
    and a histogram counted from the recorded actions.  (Round 5: with the aten statistics inside the captured graph the
    histogram came out as [600, 600, 600, 600] from the third batch on while the reward sums stayed right.)"""
    env, model, tr = build(agent, 4096, True)
    tr.evaluate(n_envs=64)
    for b in range(6):
        tr.run_batch()
        m, s, c = tr.evaluate(n_envs=64)
        ev = list(tr._eval_cache.values())[0]
        assert ev['graph'] is not None and len(tr._eval_cache) == 1
        got = (m, s, c, list(tr.last_eval_action_share), ev['hist'].clone(), ev['total'].clone(), ev['steps'].clone(), ev['acts'].clone())
        acts, D = ev['acts'], ev['done'].double()
        T, n = D.shape
        alive = torch.cat([torch.ones(1, n, dtype=torch.float64, device=D.device), torch.cumprod(1.0 - D, dim=0)[:-1]], dim=0)
        counted = torch.stack([((acts == k).double() * alive.view(T, n, 1)).sum() for k in range(model.n_a)])
        assert torch.equal(got[4], counted), 'batch %d: histogram %s, counted %s' % (b + 1, got[4].tolist(), counted.tolist())
        assert got[4].sum().item() == alive.sum().item() * env.n_agent
        ev['prepare'](); ev['episode'](); ev['statistics']()            # the same episode as eager launches
        torch.cuda.synchronize()
        assert torch.equal(got[7], ev['acts']) and torch.equal(got[5], ev['total']) and torch.equal(got[6], ev['steps'])
        assert torch.equal(got[4], ev['hist'])


def test_saved_activations_equal_recomputed_forward():
    """Uncoupled nets: the update fed by the rollout's saved activations == the update that recomputes its forward pass
    (3 batches at E = 4096: weights, values, returns)."""
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    out = []
    for save in (True, False):
        cp = cacc_config(agent='ia2c_fp', scenario='catchup', n_step=60, reward_norm=800.0)
        env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=4096)
        np.random.seed(12)
        model = models.IA2C_FP(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                               cp['MODEL_CONFIG'], seed=12, num_envs=4096)
        tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True, save_activations=save)
        assert tr.saved_acts == save
        for _ in range(3):
            tr.run_batch()
        torch.cuda.synchronize()
        out.append((model.policy.params.flat.clone(), model.buf_v.clone(), model.R.clone(), model.buf_act.clone()))
        del env, model, tr
    assert torch.equal(out[0][3], out[1][3])
    torch.testing.assert_close(out[0][1], out[1][1], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out[0][2], out[1][2], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc', 'ma2c_ic3'])
def test_compact_observation_equals_gathered_slab(agent, monkeypatch):
    """CACC: the env writing each vehicle's own 5 features ([E,8,5], the encoder gathers the neighbours inside its
    kernel, update() expands the batch once) == the env writing the pre-gathered [E,8,15] slab: same actions, values,
    weights after 3 batches.  (Encoder launches on both sides: the in-kernel encoders of IA2C-FP add their products in the
    matrix cores' order, see test_inkernel_encoders_equal_the_encoder_launch.)"""
    monkeypatch.setenv('NMARL_INKERNEL_ENCODE', '0')
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    out = []
    for compact in (True, False):
        cp = cacc_config(agent=agent, scenario='slowdown', n_step=60, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
        env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=1024)
        np.random.seed(12)
        cls = {'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3}[agent]
        model = cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                    cp['MODEL_CONFIG'], seed=12, num_envs=1024)
        tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True, compact_obs=compact)
        assert tr.compact_obs == compact and model.buf_x.shape[-1] == (5 if compact else 15)
        for _ in range(3):
            tr.run_batch()
        torch.cuda.synchronize()
        out.append((model.policy.params.flat.clone(), model.buf_v.clone(), model.buf_act.clone(), env.h.clone()))
        del env, model, tr
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][3], out[1][3])
    torch.testing.assert_close(out[0][1], out[1][1], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('E', [4096, 1000, 77])
@pytest.mark.parametrize('agent,scenario', [('ia2c_fp', 'catchup'), ('ia2c_fp', 'slowdown'), ('ma2c_nc', 'catchup'), ('ma2c_nc', 'slowdown'),
                                            ('ia2c', 'catchup'), ('ma2c_cu', 'slowdown')])
def test_env_step_inside_the_lock_step_launch_is_the_env_kernel(E, agent, scenario, monkeypatch):
    """ONE launch per lock-step (IA2C-FP: lstm_step_x_kernel<3,0,1>; round 6: NeurComm <4,1,1>, IA2C / ConseNet <3,0,2>): the lock-step kernel stepping the
    env itself behind its action draw (ENV block: the last of the 8 agents' waves that own a strip of 16 replicas steps them, no
    wave waits) against the same kernel followed by the env kernel (nmarl_cacc_step) -- the same device function on the same
    actions, so EVERYTHING is bit-identical after 3 batches through the hipGraph: actions, rewards, done flags, observations,
    env state incl. the fused auto-reset at the episode end (T = 3 batches here), values, weights.  E = 1000 / 77: ragged last
    row block (strips with fewer than 16 replicas, waves with none)."""
    monkeypatch.setenv('NMARL_NC_ENV_IN_KERNEL', '1')      # (NeurComm's default keeps the env kernel: measured no faster inside)
    out = []
    for inside in ('1', '0'):
        monkeypatch.setenv('NMARL_INKERNEL_ENV', inside)
        env, model, tr = build(agent, E, True, scenario=scenario, n_step=20, episode_sec=6)      # T = 60 lock-steps = 3 batches
        assert tr.enc_in_kernel and tr.env_in_kernel == (inside == '1') and not tr.fused_encode
        rec = []
        for _ in range(3):
            tr.run_batch()
            rec += [model.buf_act.clone(), tr.buf_rraw.clone(), tr.buf_g.clone(), model.buf_done_post.clone(), model.buf_x.clone()]
        tr.flush()
        torch.cuda.synchronize()
        assert tr.handoff_fallbacks == 0
        assert int(env.episode.min()) == 2 and int(env.t.max()) == 0          # every replica finished an episode and was re-initialised
        out.append(rec + [env.h.clone(), env.v.clone(), env.u.clone(), env.t.clone(), env.collided.clone(), env.v0_init.clone(),
                          env.episode.clone(), model.buf_v.clone(), model.policy.params.flat.clone(), tr.ep_sum.clone()])
        st = tr.stats()
        assert st['episodes'] == E
        del env, model, tr
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), 'item %d differs between the in-launch env step and the env kernel' % k


@pytest.mark.parametrize('agent,scenario,E', [('ia2c_fp', 'catchup', 4096), ('ia2c', 'catchup', 1000), ('ma2c_cu', 'slowdown', 77), ('ma2c_nc', 'slowdown', 1024),
                                              ('ma2c_ic3', 'grid', 256), ('ma2c_dial', 'catchup', 512)])
def test_fused_heads_loss_pass_equals_the_autograd_chain(agent, scenario, E, monkeypatch):
    """The update's heads + A2C loss + heads' backward as ONE pass over the h sequence (nmarl_heads_loss, round 6) against the chain it
    replaces (skinny GEMM -> nmarl_a2c_loss_fwd / _bwd -> nmarl_thin_linear_bwd; policies.py:20-30, 50-77): same rollout (bit-identical
    actions), the loss terms and the flat gradient of the first update at fp32 summation-order tolerance, the weights after 3 batches
    close.  (The two differ in the order the 64-long head dots and the weight-gradient partial sums are added.)"""
    out = []
    for fused in ('1', '0', 'dh'):            # 'dh': the fused pass writes dL/dh as a tensor, the one-launch BPTT does not expand dy8 itself
        monkeypatch.setenv('NMARL_FUSED_HEADS_LOSS', '0' if fused == '0' else '1')
        monkeypatch.setenv('NMARL_BPTT_HEAD_DY', '0' if fused == 'dh' else '1')
        env, model, tr = build(agent, E, False, scenario=scenario, n_step=20)
        tr.run_batch()
        torch.cuda.synchronize()
        rec = [model.buf_act.clone(), torch.stack([t.detach().clone() for t in model.last_loss[:3]]), model.policy.params.grad.clone()]
        for _ in range(2):
            tr.run_batch()
        tr.flush()
        torch.cuda.synchronize()
        out.append(rec + [model.policy.params.flat.clone()])
        del env, model, tr
    (act1, loss1, g1, w1), (act0, loss0, g0, w0), (act2, loss2, g2, w2) = out
    assert torch.equal(act1, act0) and torch.equal(act2, act0) and torch.equal(loss1, loss2)
    torch.testing.assert_close(loss1, loss0, rtol=1e-5, atol=1e-7)
    scale = float(g0.abs().max())
    for g, w in ((g1, w1), (g2, w2)):
        torch.testing.assert_close(g, g0, rtol=1e-4, atol=2e-6 * scale)
        torch.testing.assert_close(w, w0, rtol=1e-3, atol=1e-5)


def test_a_recurrence_that_ignores_the_heads_dy8_fails_loudly(monkeypatch):
    """The fused update hands the heads' dL/dh to the recurrence as dy8 behind a zero-stride placeholder gradient
    (ops.head_dy_placeholder): a backward that does not take it (ops.take_head_dy) would silently train on a zero gradient -- the
    model checks that it was consumed and raises."""
    from deeprl_network_amd import ops
    env, model, tr = build('ia2c_fp', 64, False, n_step=10)
    tr.rollout()
    model.load_rewards(tr.buf_rraw)
    monkeypatch.setattr(ops, 'take_head_dy', lambda dHs: None)
    with pytest.raises(RuntimeError, match='did not take'):
        model.update_grads(tr.R_end)


@pytest.mark.parametrize('use_graph', [True, False])
@pytest.mark.parametrize('agent,scenario,E', [('ma2c_nc', 'slowdown', 4096), ('ma2c_nc', 'catchup', 1000), ('ma2c_ic3', 'slowdown', 77),
                                              ('ma2c_ic3', 'grid', 1024), ('ma2c_ic3', 'grid', 1000)])
def test_message_term_handed_from_the_re_step_to_the_next_lock_step(agent, scenario, E, use_graph, monkeypatch):
    """Coupled nets, one launch per lock-step (round 6): the value re-step of lock-step t computes its message term from the
    neighbours' new, un-masked h (quirk Q3, agents/utils.py:182-199 / 395-400) -- which is exactly the policy step's message term of
    lock-step t + 1 (utils.py:129-149 called again on the states t left).  The launch hands it on (and CommNet's mean rows, the
    update's message-weight-gradient input) instead of the next launch recomputing it: kernels <4,.,.,1> (t = 0) and <4,.,.,2>
    (t >= 1, bootstrap) against <4,.,.,0> everywhere -- EVERYTHING bit-identical after 3 batches: actions, values, saved LSTM
    inputs / means (through the weights the update makes of them), env state, weights, optimiser slots."""
    out = []
    for carry in ('1', '0'):
        monkeypatch.setenv('NMARL_MSG_CARRY', carry)
        env, model, tr = build(agent, E, use_graph, scenario=scenario, n_step=20)
        assert model.policy.pv_one_launch(E)
        calls = {'in': 0, 'out': 0}
        orig = type(model)._msg_carry

        def spy(self, t, orig=orig, calls=calls):
            d = orig(self, t)
            if d:
                calls['in'] += d.get('carry_in') is not None
                calls['out'] += d.get('carry_out') is not None
            return d
        monkeypatch.setattr(type(model), '_msg_carry', spy)
        rec = []
        for _ in range(3):
            tr.run_batch()
            rec += [model.buf_act.clone(), model.buf_vn.clone(), model.S_buf.clone()] + [v.clone() for v in model.policy._extra_full.values()]
        tr.flush()
        torch.cuda.synchronize()
        monkeypatch.setattr(type(model), '_msg_carry', orig)
        assert tr.handoff_fallbacks == 0
        # (hipGraph: Python runs for the warm-up and the capture only; eager: every batch) lock-steps 1..20 start from a carried term
        assert (calls['in'] > 0 and calls['in'] % 20 == 0 and calls['out'] == calls['in']) if carry == '1' else calls['in'] == calls['out'] == 0
        if 'MM' in model.policy._extra_full:
            assert float(model.policy._extra_full['MM'][:, -1].abs().max()) == 0.0       # the padding slab stays zero
        out.append(rec + [t.clone() for t in env.state_tensors()] + [model.policy.params.flat.clone(), model.policy.params.ms.clone(), tr.R_end.clone()])
        del env, model, tr
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), 'item %d differs between the carried and the recomputed message term' % k


@pytest.mark.parametrize('use_graph', [True, False])
@pytest.mark.parametrize('E', [1024, 1000, 77, 7000])
def test_grid_env_step_as_a_role_of_the_lock_step_launch_is_the_env_kernel(E, use_graph, monkeypatch):
    """BASELINE configs[3] (synthetic 5 x 5 grid, CommNet): the env step run by extra blocks of the one-launch lock-step
    (lstm_step_x_kernel<4,2,0>, GENV role on the compute units the 25 x ceil(E / 128) LSTM blocks leave idle: they take the drawn
    actions from two hand-off words per replica) against the same launch followed by nmarl_grid_step -- one device function
    (csrc/grid_tile.h) on the same actions, so everything is bit-identical after 3 batches: actions, rewards, done flags,
    observations, env state incl. the auto-reset at the episode end (T = 3 batches), values, weights; the hand-off words are
    left zero.  E = 1000 / 77: ragged last row block and last group of 16 replicas; E = 7000: more LSTM blocks (1375) than the chip
    holds -> no one-launch lock-step at all, the form must be refused, not mis-selected."""
    from deeprl_network_amd import ops
    out = []
    for inside in ('1', '0'):
        monkeypatch.setenv('NMARL_GRID_ENV_IN_KERNEL', inside)
        env, model, tr = build('ma2c_ic3', E, use_graph, scenario='grid', n_step=20, episode_sec=300)      # T = 60 lock-steps = 3 batches
        assert env.T == 60
        fits = ops.step_grid_env_blocks(25, E) > 0 and model.policy.pv_one_launch(E)
        assert fits == (E < 7000)
        assert tr.env_in_kernel == (inside == '1' and fits)
        rec = []
        for _ in range(3):
            tr.run_batch()
            rec += [model.buf_act.clone(), tr.buf_rraw.clone(), tr.buf_g.clone(), model.buf_done_post.clone(), model.buf_x.clone()]
        tr.flush()
        torch.cuda.synchronize()
        assert tr.handoff_fallbacks == 0
        assert int(env.episode.min()) == 2 and int(env.t.max()) == 0          # every replica finished an episode and was re-initialised
        if tr.env_in_kernel:
            assert int(env._words.abs().max()) == 0
        out.append(rec + [t.clone() for t in env.state_tensors()] + [model.buf_v.clone(), model.policy.params.flat.clone(), tr.ep_sum.clone()])
        assert tr.stats()['episodes'] == E
        del env, model, tr
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), 'item %d differs between the in-launch env step and the env kernel' % k


@pytest.mark.parametrize('E', [4096, 1000, 77])
@pytest.mark.parametrize('agent,scenario', [('ia2c_fp', 'catchup'), ('ia2c_fp', 'slowdown'), ('ma2c_nc', 'slowdown'), ('ia2c', 'catchup')])
def test_env_step_inside_the_lock_step_launch_vs_oracle(E, agent, scenario, monkeypatch):
    """The env step INSIDE the lock-step launch (lstm_step_x_kernel<3,0,1> / NeurComm's <4,1,1> + ENV block) against oracle/cacc_ref.py directly
    (cacc_env.py:191-242, 40-79, 166-189), not through the env kernel: every lock-step of 3 batches (= one 60-step episode,
    auto-reset at its end) is one launch; the env state in front of each launch is read back, the fp32 oracle is put into that
    state and stepped with the actions the launch drew, and the launch's observation, reward, global reward, done flag, new
    state and -- at the episode end -- the Philox re-initialisation are compared at the per-step tolerance of SURVEY 8c
    (rtol 1e-5; replicas whose min headway lands within 1e-4 of h_min excluded and counted)."""
    from oracle import philox
    from oracle.cacc_ref import CaccBatchRef, CaccParams
    monkeypatch.setenv('NMARL_NC_ENV_IN_KERNEL', '1')
    env, model, tr = build(agent, E, False, scenario=scenario, n_step=20, episode_sec=6)     # 60 lock-steps: the third batch ends the episode
    assert tr.enc_in_kernel and tr.env_in_kernel and not tr.fused_encode and env.T == 60
    ref = CaccBatchRef(CaccParams(config=env.config), E=E, dtype=np.float32, train_mode=True)
    assert ref.p.T == 60 and ref.p.batch_size == 20
    ref.reset(np.zeros(E, np.float32))
    episode = np.ones(E, dtype=np.int64)                      # the episode a replica's NEXT re-initialisation starts
    pre = []
    orig_act = model.act

    def act(*a, **kw):
        pre.append([t.cpu().numpy().copy() for t in (env.h, env.v, env.u, env.t, env.collided, env.v0_init)])
        return orig_act(*a, **kw)
    model.act = act
    tol = dict(rtol=1e-5, atol=1e-6)
    excluded = steps = 0
    for b in range(3):
        del pre[:]
        tr.run_batch()
        torch.cuda.synchronize()
        pre.append([t.cpu().numpy().copy() for t in (env.h, env.v, env.u, env.t, env.collided, env.v0_init)])
        acts, rraw, g = model.buf_act.cpu().numpy(), tr.buf_rraw.cpu().numpy(), tr.buf_g.cpu().numpy()
        done, X = model.buf_done_post.cpu().numpy().astype(bool), model.buf_x.cpu().numpy()
        for t in range(20):
            ref.h, ref.v, ref.u = pre[t][0].copy(), pre[t][1].copy(), pre[t][2].copy()
            ref.t, ref.collided, ref.v0_init = pre[t][3].astype(np.int64), pre[t][4].astype(bool), pre[t][5].copy()
            ro, rr, rd, rg = ref.step(acts[t])
            ok = np.abs(ref.h.min(axis=1) - 1.0) > 1e-4
            excluded += int((~ok).sum())
            steps += 1
            assert np.array_equal(done[t][ok], rd[ok]) and (bool(rd.all()) or not (b == 2 and t == 19))
            np.testing.assert_allclose(rraw[t][ok], rr[ok], rtol=1e-5, atol=1e-3)
            np.testing.assert_allclose(g[t][ok], rg[ok], rtol=1e-5, atol=1e-3)
            if t == 19 and done[t].any():                     # the fused auto-reset behind a batch's last lock-step (Q4)
                ro = ref.reset(philox.reset_uniform(env.seed, env.env_id_base + np.arange(E), episode), mask=done[t])
                episode += done[t]
            nh, nv, nu, nt, nc, nv0 = pre[t + 1]
            np.testing.assert_allclose(nh[ok], ref.h[ok], **tol)
            np.testing.assert_allclose(nv[ok], ref.v[ok], **tol)
            np.testing.assert_allclose(nu[ok], ref.u[ok], rtol=1e-5, atol=2e-5)
            np.testing.assert_allclose(nv0, ref.v0_init, **tol)
            assert np.array_equal(nt, ref.t) and np.array_equal(nc.astype(bool)[ok], ref.collided[ok])
            np.testing.assert_allclose(X[t + 1][ok], ro[ok], rtol=1e-5, atol=2e-5)
    assert steps == 60 and excluded <= max(1, E // 100)
    assert int(env.episode.min()) == 2 and int(env.t.max()) == 0


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc', 'ia2c', 'ma2c_cu'])
def test_inkernel_encoders_equal_the_encoder_launch(agent, monkeypatch):
    """IA2C-FP / NeurComm (two encoders) and IA2C / ConseNet (the observation encoder alone, <3,0,2>, round 6): the lock-step kernel
    running the input encoders itself (lstm_step_x_kernel<3,0,1> / <4,1,1>: no encoder
    launch, the env step alone behind it) against the env step + encoder launch (nmarl_cacc_step_encode) in front of the plain
    lock-step kernel, ONE batch from the same state at E = 4096 and E = 1000 (ragged last block): saved LSTM inputs, values and the
    post-update weights agree to fp32 summation order; the drawn actions are identical except where a uniform falls within
    that rounding of a CDF boundary.  (NeurComm: the encoders' output reaches the K loop through the S slot in global memory --
    a stale read-back would show as a gross mismatch between the saved inputs and everything computed from them.)"""
    for E in (4096, 1000):
        out = []
        for inside in ('1', '0'):
            monkeypatch.setenv('NMARL_INKERNEL_ENCODE', inside)
            env, model, tr = build(agent, E, True, scenario='catchup' if agent.startswith('ia2c') else 'slowdown', n_step=60)
            # (ConseNet's 5-input encoder has no fused step + encode form: its round-5 lock-step is encoder launch + step kernel + env kernel)
            assert tr.enc_in_kernel == (inside == '1') and tr.fused_encode == (inside == '0' and agent != 'ma2c_cu')
            tr.rollout()
            torch.cuda.synchronize()
            S, acts, vals = model.S_buf.clone(), model.buf_act.clone(), model.buf_vn.clone()
            G = model.G_buf.clone()
            tr._update()
            tr.flush()
            torch.cuda.synchronize()
            assert tr.handoff_fallbacks == 0
            out.append((S, acts, vals, model.policy.params.flat.clone(), G))
            del env, model, tr
        same = (out[0][1] == out[1][1]).all(dim=0).all(dim=-1)            # replicas whose whole action tape agrees
        assert same.float().mean().item() > 0.999
        torch.testing.assert_close(out[0][0][:, 0], out[1][0][:, 0], rtol=1e-5, atol=1e-6)      # first lock-step: same inputs
        torch.testing.assert_close(out[0][4][:, 0], out[1][4][:, 0], rtol=1e-4, atol=1e-5)      # ... and the gates computed from them
        torch.testing.assert_close(out[0][0][:, :, same], out[1][0][:, :, same], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(out[0][2][:, :, same], out[1][2][:, :, same], rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(out[0][3], out[1][3], rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc', 'ia2c'])
def test_fused_env_encode_equals_separate_launches(agent, monkeypatch):
    """The env kernel running the next lock-step's input encoders behind its step (nmarl_cacc_step_encode: the observation
    is encoded before it leaves the CU) == env step + nmarl_fc_fwd_multi as two launches: bit-identical actions, values,
    saved LSTM inputs and weights after 3 batches (E = 1000: a ragged last block of 16 replicas)."""
    monkeypatch.setenv('NMARL_INKERNEL_ENCODE', '0')     # (IA2C-FP otherwise runs its encoders inside the lock-step kernel)
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    out = []
    for fused in (True, False):
        cp = cacc_config(agent=agent, scenario='slowdown', n_step=60, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
        env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=1000)
        np.random.seed(12)
        cls = {'ia2c': models.IA2C, 'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC}[agent]
        model = cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                    cp['MODEL_CONFIG'], seed=12, num_envs=1000)
        tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True, fused_encode=fused)
        assert tr.fused_encode == fused
        for _ in range(3):
            tr.run_batch()
        torch.cuda.synchronize()
        out.append((model.policy.params.flat.clone(), model.buf_v.clone(), model.buf_act.clone(), env.h.clone(),
                    model.S_buf.clone()))
        del env, model, tr
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('agent,E', [('ma2c_nc', 1000), ('ma2c_ic3', 300), ('ma2c_nc', 4200)])
def test_coupled_one_launch_step_equals_two_launches(agent, E, monkeypatch):
    """Coupled nets: policy step + value re-step in ONE launch (blocks hand the new h over inside it, lstm_step_x_kernel<4,.>)
    vs the policy-step / value-step pair (NMARL_INKERNEL_HANDOFF=0) over one batch through the hipGraph: the same actions,
    saved activations and states bit for bit (the policy halves are the same arithmetic), values and updated weights equal up to
    the order of the re-step's sums.  E = 1000 / 300: ragged last blocks; E = 4200: 8 x 33 blocks > compute units -- the
    engine must fall back to the two launches by itself."""
    from deeprl_network_amd import ops
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    monkeypatch.setenv('NMARL_NC_ONE_LAUNCH', '0')          # (the encoders stay behind the env kernel in both arms: this test is about the hand-off)
    out = []
    for one in (True, False):
        if not one:
            monkeypatch.setenv('NMARL_INKERNEL_HANDOFF', '0')
        cp = cacc_config(agent=agent, scenario='slowdown', n_step=20, reward_norm=5000.0)
        env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
        np.random.seed(12)
        cls = {'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3}[agent]
        model = cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                    cp['MODEL_CONFIG'], seed=12, num_envs=E)
        tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
        tr.run_batch()
        torch.cuda.synchronize()
        ops.check_coupled_status()
        cus = torch.cuda.get_device_properties(0).multi_processor_count        # 256 on an un-partitioned MI355X
        assert model.policy.pv_one_launch(E) == (one and 8 * -(-E // 128) <= cus)
        out.append((model.buf_act.clone(), model.S_buf.clone(), model.G_buf.clone(), model.H_all.clone(), model.C_all.clone(),
                    model.buf_v.clone(), model.policy.params.flat.clone()))
        del env, model, tr
    for a, b in zip(out[0][:5], out[1][:5]):
        assert torch.equal(a, b)
    torch.testing.assert_close(out[0][5], out[1][5], rtol=2e-5, atol=5e-6)
    torch.testing.assert_close(out[0][6], out[1][6], rtol=1e-4, atol=2e-6)


@pytest.fixture
def handoff_switch():
    """The recovery pins the launch-per-step kernels process-wide: give the following tests the one-launch forms back."""
    from deeprl_network_amd import _lib, ops
    yield
    ops._handoff_off[0] = False
    _lib.lib.nmarl_test_handoff_fault(0)
    for st in ops._handoff_status.values():
        st.zero_()


@pytest.mark.parametrize('agent,site,scenario', [('ma2c_nc', 'step', 'slowdown'), ('ma2c_nc', 'bptt', 'slowdown'), ('ma2c_ic3', 'step', 'slowdown'),
                                                 ('ma2c_ic3', 'step', 'grid')])
@pytest.mark.parametrize('use_graph', [False, True])
def test_handoff_timeout_fails_closed(agent, site, scenario, use_graph, handoff_switch, monkeypatch):
    """An in-launch hand-off whose neighbour block never shows up (injected: block 0 of one launch publishes nothing, 4096
    spins) must fail CLOSED, and without the host looking at the device between batches: the poisoned batch and the batch
    launched behind it change no weight, no optimiser slot, no episode statistic and hand nothing over (the guarded RMSProp
    and the guarded batch epilogue refuse on the device, the start-of-batch snapshot is not overwritten); the trainer notices
    one batch LATE (pinned non-blocking copy of the status word), rewinds, pins the launch-per-step kernels and re-runs both
    batches -- after them weights, optimiser slots, env state, actions, returns, lr schedule and counters equal, bit for bit,
    those of a run that never used the one-launch kernels.  site: the fault hits the lock-step kernel of the rollout's first
    step / the coupled BPTT launch of the update.  scenario 'grid': the launch also steps the env (GENV role) -- block 0's
    actions never arrive either, the env blocks of its 128 replicas time out as well, and the recovery clears their words.  use_graph: the fault is armed for an eager first batch either way (the
    fault flag is a launch argument), the second batch is a graph replay or eager."""
    from deeprl_network_amd import _lib, ops
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    E, T = 256, 10
    # reference: launch-per-step kernels from the start
    monkeypatch.setenv('NMARL_INKERNEL_HANDOFF', '0')
    env, model, tr = build(agent, E, False, scenario=scenario, n_step=T)
    assert not tr.handoff_guard
    for _ in range(3):
        tr.run_batch()
    torch.cuda.synchronize()
    ref = (model.policy.params.flat.clone(), model.policy.params.ms.clone(), env.state_tensors()[0].clone(), model.buf_act.clone(), tr.R_end.clone(),
           tr.ep_sum.clone(), model.h_bw.clone(), model.buf_x[0].clone(), model.lr_scheduler.n)
    del env, model, tr
    monkeypatch.delenv('NMARL_INKERNEL_HANDOFF')
    env, model, tr = build(agent, E, use_graph, scenario=scenario, n_step=T)
    tr.global_counter = Counter(10 ** 12, 10 ** 12, 10 ** 12)
    model.policy.refresh_wimage()              # (the message image decides whether the one-launch step exists)
    assert tr.handoff_guard and model.policy.pv_one_launch(E) and tr.env_in_kernel == (scenario == 'grid')
    w0, ms0 = model.policy.params.flat.clone(), model.policy.params.ms.clone()
    seen = {}
    orig = BatchedTrainer._recover_from_handoff_timeout

    def spy(self, batches=1):
        torch.cuda.synchronize()
        seen['w'], seen['ms'] = self.model.policy.params.flat.clone(), self.model.policy.params.ms.clone()
        seen['skipped'], seen['batches'] = ops.handoff_skipped_updates(self.device), batches
        seen['ep_len'], seen['done_pre'] = self.ep_len.clone(), self.done_pre.clone()
        orig(self, batches)
    monkeypatch.setattr(BatchedTrainer, '_recover_from_handoff_timeout', spy)
    skipped0 = ops.handoff_skipped_updates('cuda')
    if use_graph:                              # the graphs are captured in front of the fault: a clean eager warm-up cannot be had
        tr.use_graph = False                   # (the fault counter counts launches), so the poisoned batch itself runs eagerly
    # nth hand-off launch from now: 1 = the first lock-step; T + 2 = the BPTT behind T lock-steps + the bootstrap step
    _lib.check(_lib.lib.nmarl_test_handoff_fault(1 if site == 'step' else T + 2), 'nmarl_test_handoff_fault')
    tr.run_batch()
    tr.use_graph = use_graph
    assert tr.handoff_fallbacks == 0 and not seen, 'the host looked at the status word of the batch it had just launched'
    tr.run_batch()                             # launched behind the poisoned one; its probe finds the word raised
    assert tr.handoff_fallbacks == 1 and not ops.handoff_enabled() and not model.policy.pv_one_launch(E) and not tr.handoff_guard
    assert not tr.env_in_kernel and (scenario != 'grid' or int(env._words.abs().max()) == 0)
    assert seen['batches'] == 2 and seen['skipped'] == skipped0 + 2
    assert torch.equal(seen['w'], w0) and torch.equal(seen['ms'], ms0), 'a refused batch reached the weights'
    assert float(seen['ep_len'].abs().max()) == 0.0 and bool((seen['done_pre'] == 1).all()), 'a refused batch was handed over'
    assert tr.n_batches == 2 and tr.global_counter.cur_step == 2 * T
    tr.run_batch()
    tr.flush()
    torch.cuda.synchronize()
    ops.check_coupled_status()
    got = (model.policy.params.flat, model.policy.params.ms, env.state_tensors()[0], model.buf_act, tr.R_end, tr.ep_sum, model.h_bw, model.buf_x[0],
           model.lr_scheduler.n)
    for name, a, b in zip(('weights', 'rmsprop slots', 'env state', 'actions', 'R_end', 'episode sums', 'h_bw', 'x_0'), got, ref):
        assert torch.equal(a, b), '%s differ from the launch-per-step run' % name
    assert got[8] == ref[8] and tr.n_batches == 3
    assert tr.stats()['episodes'] == 0


def test_handoff_timeout_in_the_last_batch_is_found_by_flush(handoff_switch):
    """The batch launched last has no batch behind it whose probe would look at its status word: `flush()` (called by
    `stats()` and at the end of `run()`) does, and re-runs that one batch."""
    from deeprl_network_amd import _lib, ops
    E, T = 256, 10
    env, model, tr = build('ma2c_nc', E, False, scenario='slowdown', n_step=T)
    model.policy.refresh_wimage()
    assert tr.handoff_guard
    tr.run_batch()
    w1 = model.policy.params.flat.clone()
    _lib.check(_lib.lib.nmarl_test_handoff_fault(1), 'nmarl_test_handoff_fault')
    tr.run_batch()
    torch.cuda.synchronize()
    assert tr.handoff_fallbacks == 0 and torch.equal(model.policy.params.flat, w1)
    st = tr.stats()
    assert tr.handoff_fallbacks == 1 and tr.n_batches == 2 and not torch.equal(model.policy.params.flat, w1) and st['episodes'] == 0
    ops.check_coupled_status()


def test_rearm_arms_the_guard_whenever_the_handoff_kernels_come_back(handoff_switch, monkeypatch):
    """After a time-out and `rearm_after` clean batches the one-launch kernels are selected again -- and with them the guard,
    also where the lock-step itself has no one-launch form at this size (the coupled BPTT still picks its hand-off form from
    the process-wide switch): a second time-out must be found and recovered from, not stall the training."""
    from deeprl_network_amd import _lib, ops
    E, T = 256, 10
    env, model, tr = build('ma2c_nc', E, False, scenario='slowdown', n_step=T, rearm_after=2)
    model.policy.refresh_wimage()
    monkeypatch.setattr(type(model.policy), 'pv_one_launch', lambda self, E_: False)      # two-launch lock-step, hand-off BPTT
    tr._select_lock_step_form()
    assert tr.handoff_guard and not tr.enc_in_kernel and tr.fused_encode
    _lib.check(_lib.lib.nmarl_test_handoff_fault(1), 'nmarl_test_handoff_fault')          # the first hand-off launch: the BPTT
    tr.run_batch()
    tr.run_batch()
    assert tr.handoff_fallbacks == 1 and not tr.handoff_guard and not ops.handoff_enabled()
    for _ in range(2):
        tr.run_batch()
    assert ops.handoff_enabled() and tr.handoff_guard, 're-armed kernels without the guard'
    w = model.policy.params.flat.clone()
    _lib.check(_lib.lib.nmarl_test_handoff_fault(1), 'nmarl_test_handoff_fault')
    tr.run_batch()
    tr.run_batch()
    assert tr.handoff_fallbacks == 2
    tr.run_batch()
    tr.flush()
    torch.cuda.synchronize()
    ops.check_coupled_status()
    assert not torch.equal(model.policy.params.flat, w) and torch.isfinite(model.policy.params.flat).all()
    assert ops.handoff_skipped_updates('cuda') >= 4


def test_handoff_capacity_and_fake_cus(monkeypatch, handoff_switch):
    """Residency comes from the occupancy API x compute units; NMARL_TEST_FAKE_CUS shrinks the device: the engine must pick
    the two-launch lock-step / step-wise BPTT by itself, and the launchers must refuse an over-sized one-launch grid."""
    from deeprl_network_amd import _lib, ops
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for which, K in ((1, 128), (1, 64), (2, 128), (2, 64)):
        assert _lib.lib.nmarl_handoff_capacity(which, K) == cus          # one 512-thread, > 80-KB-LDS block per compute unit
    assert ops.step_handoff_supported(8, 4096, 'cuda') and not ops.step_handoff_supported(8, 4096 + 128, 'cuda')
    monkeypatch.setenv('NMARL_TEST_FAKE_CUS', '16')
    assert _lib.lib.nmarl_handoff_capacity(1, 128) == 16 and _lib.lib.nmarl_handoff_capacity(2, 64) == 16
    assert ops.step_handoff_supported(8, 256, 'cuda') and not ops.step_handoff_supported(8, 257, 'cuda')
    env, model, tr = build('ma2c_nc', 512, True, scenario='slowdown', n_step=10)        # 8 x 4 blocks > 16 "compute units"
    assert not model.policy.pv_one_launch(512)
    tr.run_batch()
    torch.cuda.synchronize()
    ops.check_coupled_status()
    assert tr.handoff_fallbacks == 0 and torch.isfinite(model.policy.params.flat).all()
