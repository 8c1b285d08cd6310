"""Batched trainer on the GPU at the BASELINE size (8 agents x 4096 replicas): size-independent properties of the
whole rollout + update path -- hipGraph replay == eager launch, run-to-run determinism (no atomics anywhere),
batch invariance of the rollout (replica e of 4096 == the same replica in a batch of 16), finite updates."""
import numpy as np
import pytest
import torch

from helpers import cacc_config

pytestmark = pytest.mark.gpu


def build(agent, E, use_graph, env_id_base=0, scenario='catchup', n_step=60):
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    cp = cacc_config(agent=agent, scenario=scenario, n_step=n_step, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
    env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E, env_id_base=env_id_base)
    cls = {'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3, 'ia2c': models.IA2C,
           'ma2c_cu': models.IA2C_CU, 'ma2c_dial': models.MA2C_DIAL}[agent]
    np.random.seed(12)
    model = cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                cp['MODEL_CONFIG'], seed=12, num_envs=E)
    return env, model, BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=use_graph)


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc'])
def test_graph_equals_eager_and_is_deterministic(agent):
    E = 4096
    runs = []
    for use_graph in (True, False, True):
        env, model, tr = build(agent, E, use_graph)
        for _ in range(3):
            tr.run_batch()
        torch.cuda.synchronize()
        runs.append((model.policy.params.flat.clone(), env.h.clone(), model.buf_act.clone(), tr.R_end.clone()))
        del env, model, tr
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b), 'hipGraph replay differs from eager launches'
    for a, b in zip(runs[0], runs[2]):
        assert torch.equal(a, b), 'two identical runs differ (non-deterministic kernel?)'
    assert torch.isfinite(runs[0][0]).all()


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc', 'ma2c_ic3', 'ma2c_cu', 'ma2c_dial'])
def test_captured_update_equals_eager_update(agent, monkeypatch):
    """The A2C update replayed as a hipGraph (from the second batch on; rewards, return scan, loss, backward, clip + RMSProp,
    and for nets without the hand-off guard the batch epilogue) against the eager update behind the same rollout graph: weights,
    RMSProp slots, env state, actions, episode statistics bit-identical after 4 batches -- with a LINEAR lr schedule, whose
    value reaches the captured optimiser step through a device scalar."""
    E = 1024
    runs = []
    for capture in ('1', '0'):
        monkeypatch.setenv('NMARL_CAPTURE_UPDATE', capture)
        env, model, tr = build(agent, E, True, n_step=20)
        model.lr_scheduler = type(model.lr_scheduler)(5e-4, 1e-5, 200, decay='linear')
        for _ in range(4):
            tr.run_batch()
        torch.cuda.synchronize()
        assert (tr._upd is not None) == (capture == '1'), tr.update_capture_error
        assert tr.update_capture_error is None
        runs.append((model.policy.params.flat.clone(), model.policy.params.ms.clone(), env.h.clone(), model.buf_act.clone(),
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
    bit-identical after 4 batches at the BASELINE size (hipGraph rollout and update)."""
    runs = []
    for pair in ('1', '0'):
        monkeypatch.setenv('NMARL_FC_BWD_PAIR', pair)
        env, model, tr = build(agent, 4096, True)
        for _ in range(4):
            tr.run_batch()
        torch.cuda.synchronize()
        assert (model.S_bits is not None) == (pair == '1' and agent == 'ia2c_fp')
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
@pytest.mark.parametrize('scenario', ['catchup', 'slowdown'])
def test_env_step_inside_the_lock_step_launch_is_the_env_kernel(E, scenario, monkeypatch):
    """ONE launch per lock-step (IA2C-FP on CACC): the lock-step kernel stepping the env itself behind its action draw
    (lstm_step_x_kernel<3,0,1> with the ENV block: the last of the 8 agents' waves that own a strip of 16 replicas steps them,
    no wave waits) against the same kernel followed by the env kernel (nmarl_cacc_step) -- the same device function on the same
    actions, so EVERYTHING is bit-identical after 3 batches through the hipGraph: actions, rewards, done flags, observations,
    env state incl. the fused auto-reset at the episode end (T = 3 batches here), values, weights.  E = 1000 / 77: ragged last
    row block (strips with fewer than 16 replicas, waves with none)."""
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    out = []
    for inside in ('1', '0'):
        monkeypatch.setenv('NMARL_INKERNEL_ENV', inside)
        cp = cacc_config(agent='ia2c_fp', scenario=scenario, n_step=20, reward_norm=800.0)
        cp['ENV_CONFIG']['episode_length_sec'] = '6'                 # T = 60 lock-steps = 3 batches: the third one ends the episodes
        env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
        np.random.seed(12)
        model = models.IA2C_FP(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                               cp['MODEL_CONFIG'], seed=12, num_envs=E)
        tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
        assert tr.enc_in_kernel and tr.env_in_kernel == (inside == '1') and not tr.fused_encode
        rec = []
        for _ in range(3):
            tr.run_batch()
            rec += [model.buf_act.clone(), tr.buf_rraw.clone(), tr.buf_g.clone(), model.buf_done_post.clone(), model.buf_x.clone()]
        torch.cuda.synchronize()
        assert int(env.episode.min()) == 2 and int(env.t.max()) == 0          # every replica finished an episode and was re-initialised
        out.append(rec + [env.h.clone(), env.v.clone(), env.u.clone(), env.t.clone(), env.collided.clone(), env.v0_init.clone(),
                          env.episode.clone(), model.buf_v.clone(), model.policy.params.flat.clone(), tr.ep_sum.clone()])
        st = tr.stats()
        assert st['episodes'] == E
        del env, model, tr
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), 'item %d differs between the in-launch env step and the env kernel' % k


def test_inkernel_encoders_equal_the_encoder_launch(monkeypatch):
    """IA2C-FP: the lock-step kernel running both input encoders itself (lstm_step_x_kernel<3,0,1>: no encoder launch, the env
    step alone behind it) against the env step + encoder launch (nmarl_cacc_step_encode) in front of the plain lock-step
    kernel, ONE batch from the same state at E = 4096 and E = 1000 (ragged last block): saved LSTM inputs, values and the
    post-update weights agree to fp32 summation order; the drawn actions are identical except where a uniform falls within
    that rounding of a CDF boundary."""
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    for E in (4096, 1000):
        out = []
        for inside in ('1', '0'):
            monkeypatch.setenv('NMARL_INKERNEL_ENCODE', inside)
            cp = cacc_config(agent='ia2c_fp', scenario='catchup', n_step=60, reward_norm=800.0)
            env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E)
            np.random.seed(12)
            model = models.IA2C_FP(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 9,
                                   cp['MODEL_CONFIG'], seed=12, num_envs=E)
            tr = BatchedTrainer(env, model, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=True)
            assert tr.enc_in_kernel == (inside == '1') and tr.fused_encode == (inside == '0')
            tr.rollout()
            torch.cuda.synchronize()
            S, acts, vals = model.S_buf.clone(), model.buf_act.clone(), model.buf_vn.clone()
            tr._update()
            torch.cuda.synchronize()
            out.append((S, acts, vals, model.policy.params.flat.clone()))
            del env, model, tr
        same = (out[0][1] == out[1][1]).all(dim=0).all(dim=-1)            # replicas whose whole action tape agrees
        assert same.float().mean().item() > 0.999
        torch.testing.assert_close(out[0][0][:, 0], out[1][0][:, 0], rtol=1e-5, atol=1e-6)      # first lock-step: same inputs
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


@pytest.mark.parametrize('agent,site', [('ma2c_nc', 'step'), ('ma2c_nc', 'bptt'), ('ma2c_ic3', 'step')])
def test_handoff_timeout_fails_closed(agent, site, handoff_switch, monkeypatch):
    """An in-launch hand-off whose neighbour block never shows up (injected: block 0 of one launch publishes nothing, 4096
    spins) must fail CLOSED: the poisoned batch changes no weight and no optimiser slot (the guarded RMSProp refuses it on the
    device and counts it), the trainer rewinds the batch, pins the launch-per-step kernels and re-runs it -- after two batches
    the weights equal, bit for bit, those of a run that never used the one-launch kernels.  site: the fault hits the
    lock-step kernel of the rollout's first step / the coupled BPTT launch of the update."""
    from deeprl_network_amd import _lib, ops
    from deeprl_network_amd.utils import BatchedTrainer
    E, T = 256, 10
    # reference: launch-per-step kernels from the start
    monkeypatch.setenv('NMARL_INKERNEL_HANDOFF', '0')
    env, model, tr = build(agent, E, False, scenario='slowdown', n_step=T)
    assert not tr.handoff_guard
    for _ in range(2):
        tr.run_batch()
    torch.cuda.synchronize()
    ref = (model.policy.params.flat.clone(), model.policy.params.ms.clone(), env.h.clone(), model.buf_act.clone(), tr.R_end.clone(),
           model.lr_scheduler.n)
    del env, model, tr
    monkeypatch.delenv('NMARL_INKERNEL_HANDOFF')
    # faulty run: eager rollout (the fault flag is a launch argument, a captured graph would replay it)
    env, model, tr = build(agent, E, False, scenario='slowdown', n_step=T)
    model.policy.refresh_wimage()              # (the message image decides whether the one-launch step exists)
    assert tr.handoff_guard and model.policy.pv_one_launch(E)
    w0, ms0 = model.policy.params.flat.clone(), model.policy.params.ms.clone()
    seen = {}
    orig = BatchedTrainer._recover_from_handoff_timeout

    def spy(self):
        seen['w'], seen['ms'] = self.model.policy.params.flat.clone(), self.model.policy.params.ms.clone()
        seen['skipped'] = ops.handoff_skipped_updates(self.device)
        orig(self)
    monkeypatch.setattr(BatchedTrainer, '_recover_from_handoff_timeout', spy)
    skipped0 = ops.handoff_skipped_updates('cuda')
    # nth hand-off launch from now: 1 = the first lock-step; T + 2 = the BPTT behind T lock-steps + the bootstrap step
    _lib.check(_lib.lib.nmarl_test_handoff_fault(1 if site == 'step' else T + 2), 'nmarl_test_handoff_fault')
    tr.run_batch()
    assert tr.handoff_fallbacks == 1 and not ops.handoff_enabled() and not model.policy.pv_one_launch(E)
    assert torch.equal(seen['w'], w0) and torch.equal(seen['ms'], ms0), 'the poisoned batch reached the weights'
    assert seen['skipped'] == skipped0 + 1
    tr.run_batch()
    torch.cuda.synchronize()
    ops.check_coupled_status()
    got = (model.policy.params.flat, model.policy.params.ms, env.h, model.buf_act, tr.R_end, model.lr_scheduler.n)
    for name, a, b in zip(('weights', 'rmsprop slots', 'env state', 'actions', 'R_end'), got, ref):
        assert torch.equal(a, b), '%s differ from the launch-per-step run' % name
    assert got[5] == ref[5]
    assert tr.stats()['episodes'] == 0


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
