"""Host logic of the batched trainer on CPU (HIP ops and the env replaced by their oracle
restatements, tests/cpu_emulation.py): buffer slots, batch-boundary resets, bootstrap masking,
episode statistics, and E=1 equivalence of the batched engine with the reference-API path."""
import numpy as np
import pytest
import torch

from cpu_emulation import CpuCaccBatchEnv, cpu_ops
from helpers import cacc_config


def build(agent, E, n_step=10, seed=12, scenario='catchup', env_id_base=0):
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    cp = cacc_config(agent=agent, n_step=n_step, scenario=scenario, seed=seed, reward_norm=800.0)
    # shorten the episode so the test sees episode boundaries: T = 3 batches
    cp['ENV_CONFIG']['episode_length_sec'] = '3'
    env = CpuCaccBatchEnv(cp['ENV_CONFIG'], num_envs=E, env_id_base=env_id_base)
    assert env.T == 3 * n_step
    cls = {'ia2c': models.IA2C, 'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3,
           'ma2c_cu': models.IA2C_CU, 'ma2c_dial': models.MA2C_DIAL}[agent]
    np.random.seed(seed)
    model = cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 6,
                cp['MODEL_CONFIG'], seed=seed, num_envs=E, device='cpu')
    tr = BatchedTrainer(env, model, Counter(10 ** 6, 10 ** 7, 10 ** 4), use_graph=False)
    return env, model, tr


@pytest.mark.parametrize('agent', ['ia2c', 'ia2c_fp', 'ma2c_nc', 'ma2c_ic3', 'ma2c_cu', 'ma2c_dial'])
def test_batched_trainer_runs_and_learns_something(agent):
    with cpu_ops():
        env, model, tr = build(agent, E=5)
        w0 = model.policy.params.flat.clone()
        for _ in range(7):
            tr.run_batch()
        assert torch.isfinite(model.policy.params.flat).all()
        assert not torch.equal(w0, model.policy.params.flat)
        st = tr.stats()
        assert st['episodes'] == 5 * 2          # 7 batches = 2 full episodes (+1 batch) per replica
        assert np.isfinite(st['avg_reward'])
        assert tr.global_counter.cur_step == 7 * 10            # lock-steps (env steps per replica)
        assert int(tr.step_dev.item()) == 7 * 11   # n_step draws + 1 bootstrap draw per batch


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc', 'ma2c_dial'])
def test_fused_heads_branch_equals_composed_branch(agent, monkeypatch):
    """BatchedPolicy.step_policy / step_value: the one-kernel branch (H = 64, A <= 8) and the composed branch
    (step + pi + sample_actions; step + nbr_onehot + value) are the same computation."""
    from deeprl_network_amd.agents.policies import BatchedPolicy
    with cpu_ops():
        _, m_fused, t_fused = build(agent, E=3)
        assert m_fused.policy.fused_heads
        for _ in range(2):
            t_fused.run_batch()
        monkeypatch.setattr(BatchedPolicy, 'fused_heads', property(lambda self: False))
        _, m_comp, t_comp = build(agent, E=3)
        assert not m_comp.policy.fused_heads
        for _ in range(2):
            t_comp.run_batch()
        assert torch.equal(m_fused.buf_act, m_comp.buf_act)
        torch.testing.assert_close(m_fused.buf_v, m_comp.buf_v, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m_fused.buf_fp, m_comp.buf_fp, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m_fused.policy.params.flat, m_comp.policy.params.flat, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('agent', ['ma2c_nc', 'ma2c_ic3'])
def test_coupled_one_launch_branch_equals_two_launch_branch(agent, monkeypatch):
    """Coupled nets: the host code of the one-launch policy + value step (BatchedPolicy.pv_one_launch: values via the
    agent-major buffer, the critic's action term added in update(), bootstrap from slot T) and of the policy-step /
    value-step pair are the same computation."""
    with cpu_ops():
        _, m_one, t_one = build(agent, E=3)
        for _ in range(4):
            t_one.run_batch()
        assert m_one.save_acts and m_one.policy.pv_one_launch(3)
        monkeypatch.setenv('NMARL_INKERNEL_HANDOFF', '0')
        _, m_two, t_two = build(agent, E=3)
        for _ in range(4):
            t_two.run_batch()
        assert not m_two.policy.pv_one_launch(3)
        assert torch.equal(m_one.buf_act, m_two.buf_act)
        torch.testing.assert_close(m_one.buf_v, m_two.buf_v, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m_one.policy.params.flat, m_two.policy.params.flat, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('agent', ['ma2c_nc', 'ma2c_ic3'])
def test_message_term_carry_host_logic(agent, monkeypatch):
    """Host side of the round-6 carry (agents/models.py `_msg_carry`): lock-step t's value re-step hands its message term (and
    CommNet's mean rows, into slot t + 1 of the saved means) to lock-step t + 1's policy step; lock-step 0 of a batch computes its
    own, the bootstrap step takes but does not hand on, the zero padding slab of the saved means stays zero.  On the restated ops the
    carried and the recomputed term are the same numbers: everything bit-identical after 4 batches."""
    out = []
    for carry in ('1', '0'):
        monkeypatch.setenv('NMARL_MSG_CARRY', carry)
        with cpu_ops():
            _, m, t = build(agent, E=3)
            seen = []
            def spy(self, step, orig=type(m)._msg_carry, seen=seen):
                d = orig(self, step)
                seen.append((step, None if d is None else (d.get('carry_in') is not None, d.get('carry_out') is not None, d.get('mean_next') is not None)))
                return d
            monkeypatch.setattr(type(m), '_msg_carry', spy)
            for _ in range(4):
                t.run_batch()
            monkeypatch.undo()
            monkeypatch.setenv('NMARL_MSG_CARRY', carry)
            T = m.n_step
            assert m.policy.pv_one_launch(3) and len(seen) == 4 * (T + 1)
            if carry == '1':
                per_batch = seen[:T + 1]
                assert per_batch[0] == (0, (False, True, agent == 'ma2c_ic3' and T > 1))
                assert all(per_batch[k] == (k, (True, True, agent == 'ma2c_ic3' and k + 1 < T)) for k in range(1, T))
                assert per_batch[T] == (T, (True, False, False))
            else:
                assert all(d is None for _, d in seen)
            if 'MM' in m.policy._extra_full:
                assert float(m.policy._extra_full['MM'][:, -1].abs().max()) == 0.0
            out.append((m.buf_act.clone(), m.buf_v.clone(), m.policy.params.flat.clone(), {k: v.clone() for k, v in m.policy._extra_full.items()}))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    for k in out[0][3]:
        assert torch.equal(out[0][3][k], out[1][3][k]), k


def test_dial_sender_layer_in_the_policy_step_equals_fc_launches(monkeypatch):
    """lstm_dial: the policy step also runs the sender layer on the new h (msg['next']), the value re-step and the next
    lock-step's policy step re-use its output (DIALMultiAgentPolicy._cached_msg) -- ONE fc launch on h per batch (lock-step 0:
    the weights changed) instead of two per lock-step; same actions, values, saved message terms and weights as the fc path."""
    from deeprl_network_amd.agents.policies import DIALMultiAgentPolicy
    calls = {'n': 0}
    orig = DIALMultiAgentPolicy._fc_infer

    def counting(self, x, w_key, b_key, act, out=None):
        calls['n'] += w_key == 'mfc_w'
        return orig(self, x, w_key, b_key, act, out=out)
    monkeypatch.setattr(DIALMultiAgentPolicy, '_fc_infer', counting)
    with cpu_ops():
        _, m_new, t_new = build('ma2c_dial', E=3)
        for _ in range(4):
            t_new.run_batch()
        assert m_new.save_acts and m_new.policy._msg() is not None
        n_new = calls['n']
        assert n_new == 4                                   # lock-step 0 of each batch
        calls['n'] = 0
        monkeypatch.setattr(DIALMultiAgentPolicy, '_cached_msg', lambda self, h: None)
        monkeypatch.setattr(DIALMultiAgentPolicy, '_mfc_img', property(lambda self: None, lambda self, v: None))
        _, m_old, t_old = build('ma2c_dial', E=3)
        for _ in range(4):
            t_old.run_batch()
        assert calls['n'] == 4 * (2 * 10 + 2)               # policy + value step of 10 lock-steps and of the bootstrap
        assert torch.equal(m_new.buf_act, m_old.buf_act)
        torch.testing.assert_close(m_new.buf_v, m_old.buf_v, rtol=1e-5, atol=1e-6)
        for k in ('A1', 'A2'):
            torch.testing.assert_close(m_new.policy._extra[k], m_old.policy._extra[k], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m_new.policy.params.flat, m_old.policy.params.flat, rtol=1e-4, atol=1e-6)


def test_commnet_saved_neighbour_means_equal_the_averaging_pass(monkeypatch):
    """CommNet: the policy step keeps mean_nbr(h_{t-1}) (msg['mean_out']) and the update's message-weight gradient multiplies by
    that buffer; same weights as the update that averages the h sequence itself."""
    from deeprl_network_amd.agents.policies import IC3MultiAgentPolicy
    with cpu_ops():
        _, m_new, t_new = build('ma2c_ic3', E=3)
        for _ in range(4):
            t_new.run_batch()
        assert float(m_new.policy._extra['MM'].abs().max()) > 0 and float(m_new.policy._extra_full['MM'][:, -1].abs().max()) == 0.0
        monkeypatch.setattr(IC3MultiAgentPolicy, 'save_spec', lambda self: {'ENC': self.n_h})
        _, m_old, t_old = build('ma2c_ic3', E=3)
        for _ in range(4):
            t_old.run_batch()
        assert 'MM' not in m_old.policy._extra
        assert torch.equal(m_new.buf_act, m_old.buf_act)
        torch.testing.assert_close(m_new.policy.params.flat, m_old.policy.params.flat, rtol=1e-5, atol=1e-7)


def test_batch_invariance_of_rollout():
    """Replica e of an E-replica rollout == the same replica rolled out alone (same Philox ids)."""
    with cpu_ops():
        env, model, tr = build('ma2c_nc', E=4)
        tr._rollout()
        big = (model.buf_act.clone(), model.buf_v.clone(), tr.R_end.clone())
        for e in (0, 3):
            cp_env, m1, t1 = build('ma2c_nc', E=1, env_id_base=e)
            t1._rollout()
            assert torch.equal(m1.buf_act[:, 0], big[0][:, e])
            torch.testing.assert_close(m1.buf_v[:, :, 0], big[1][:, :, e], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(t1.R_end[:, 0], big[2][:, e], rtol=1e-5, atol=1e-6)


def test_finished_replicas_restart_clean():
    with cpu_ops():
        env, model, tr = build('ia2c_fp', E=3)
        for _ in range(3):
            tr.run_batch()
        # T = 3 batches: every replica just finished its first episode
        assert torch.all(tr.done_pre == 1) and torch.all(env.episode == 2)
        assert torch.all(model.h_fw == 0) and torch.all(model.c_bw == 0)
        assert torch.allclose(model.fp, torch.full_like(model.fp, 0.25))
        assert torch.all(tr.R_end == 0)
        tr.run_batch()
        assert torch.all(tr.done_pre == 0) and not torch.all(model.h_fw == 0)


@pytest.mark.parametrize('agent', ['ma2c_nc', 'ia2c_fp', 'ma2c_dial'])
def test_heterogeneous_agents_on_the_network_cpu(agent):
    """Host logic of the batched engine for the Monaco-like network (28 agents with 2..6 actions, 2..22 own features,
    0..4 neighbours) on the CPU emulation: per-agent action ranges, padded fingerprints, frozen padded parameters,
    the spatially discounted per-agent rewards with unreachable pairs (distance -1)."""
    from cpu_emulation import CpuRealNetBatchEnv
    from helpers import net_config
    from deeprl_network_amd.main import AGENTS
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    with cpu_ops():
        cp = net_config(agent=agent, n_step=6)
        cp['ENV_CONFIG']['episode_length_sec'] = '60'                 # T = 12 = 2 batches
        env = CpuRealNetBatchEnv(cp['ENV_CONFIG'], num_envs=3)
        np.random.seed(4)
        model = AGENTS[agent](env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 6,
                              cp['MODEL_CONFIG'], seed=4, num_envs=3, device='cpu', n_feat_ls=env.n_feat_ls)
        ps = model.policy.params
        w0, pad0 = ps.flat.clone(), ps['pi_b'].detach().clone()
        tr = BatchedTrainer(env, model, Counter(10 ** 6, 10 ** 7, 10 ** 4), use_graph=False)
        for _ in range(5):
            tr.run_batch()
        acts = model.buf_act.numpy()
        fp = model.buf_fp.numpy()
        for i, n in enumerate(env.n_a_ls):
            assert acts[:, :, i].max() < n
            assert (fp[:, i, :, n:] == 0).all()
            np.testing.assert_allclose(fp[:, i, :, :n].sum(-1), 1.0, rtol=1e-5)
            assert torch.equal(ps['pi_b'].detach()[i, n:], pad0[i, n:])
        assert torch.isfinite(ps.flat).all() and not torch.equal(w0, ps.flat)
        if ps.mask is not None:
            assert torch.equal(ps.flat[ps.mask == 0], w0[ps.mask == 0])
        assert tr.stats()['episodes'] == 3 * 2


class _Cpu8FeatureEnv(CpuCaccBatchEnv):
    """TEST-ONLY: the CACC stand-in with 8 features per vehicle (the 5 of the oracle + 3 derived ones): an observation with
    16-byte feature pieces like the grid's, so CommNet's one-launch step may run the observation encoder itself."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.n_s_ls = [8] * self.n_agent

    def set_compact_obs(self, flag=True):
        assert flag, 'this stand-in emits the compact observation only'
        self.compact_obs = True
        self.obs = torch.zeros(self.E, self.n_agent, 8)
        return True

    def _emit(self):
        from cpu_emulation import np_f32
        vs = torch.from_numpy(np_f32(self.ref.veh_state()))
        self.obs.copy_(torch.cat([vs, torch.tanh(vs[..., :3])], dim=-1))
        return self.obs


def test_commnet_encoder_inside_the_step_equals_separate_encoder_launch(monkeypatch):
    """The batched engine on an observation with 16-byte feature pieces: CommNet's one-launch lock-step running the observation
    encoder itself (models.act / bootstrap hand the compact observation to the step, the encoder's output lands in the ENC slots
    the update reads) == the separate encoder launch, over three batches incl. episode ends."""
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.agents.policies import IC3MultiAgentPolicy
    from deeprl_network_amd.utils import BatchedTrainer, Counter

    def run(in_step):
        if not in_step:
            monkeypatch.setattr(IC3MultiAgentPolicy, 'encodes_in_step', lambda self, E, compact: False)
        cp = cacc_config(agent='ma2c_ic3', n_step=10, scenario='catchup', seed=12, reward_norm=800.0)
        cp['ENV_CONFIG']['episode_length_sec'] = '3'
        env = _Cpu8FeatureEnv(cp['ENV_CONFIG'], num_envs=4)
        np.random.seed(12)
        model = models.MA2C_IC3(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 6,
                                cp['MODEL_CONFIG'], seed=12, num_envs=4, device='cpu')
        tr = BatchedTrainer(env, model, Counter(10 ** 6, 10 ** 7, 10 ** 4), use_graph=False)
        assert tr.compact_obs and model.buf_x.shape[-1] == 8
        for _ in range(4):
            tr.run_batch()
        assert model.policy.encodes_in_step(4, model.compact_obs) == in_step
        return model.buf_act.clone(), model.buf_v.clone(), model.policy._extra['ENC'].clone(), model.policy.params.flat.clone()

    with cpu_ops():
        a, b = run(True), run(False)
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-6)


def test_scheduler_follows_the_reference_rule_and_rewinds():
    """agents/utils.py:917-930 of the reference: `get(n)` advances the step count by n and THEN evaluates
    max(val_min, val_init * (1 - n / total_step)) for 'linear', val_init otherwise.  `rewind` gives the steps of a re-run batch
    back (hand-off time-out recovery), `at` reads without advancing (the device scalar of a captured update)."""
    from deeprl_network_amd.agents.utils import Scheduler
    s = Scheduler(5e-4, 1e-5, 1000, decay='linear')
    n, got = 0, []
    for step in (60, 60, 120, 700, 100):
        n += step
        got.append((s.get(step), max(1e-5, 5e-4 * (1 - n / 1000.0))))
    assert all(a == b for a, b in got) and got[-1][0] == 1e-5 and s.n == 1040      # clamped at val_min past total_step
    assert not s.constant and s.at(500) == 5e-4 * 0.5 and s.n == 1040             # `at` does not advance
    s.rewind(100)
    assert s.n == 940 and s.get(100) == got[-1][0]                               # the re-run batch sees the same rate
    c = Scheduler(5e-4, decay='constant')
    assert c.constant and c.get(60) == 5e-4 and c.get(10 ** 9) == 5e-4 and c.n == 60 + 10 ** 9
