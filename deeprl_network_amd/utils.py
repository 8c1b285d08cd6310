"""Rollout / training loops -- the reference's root utils.py (Counter 70-97,
Trainer 100-254, Evaluator 311-336) plus the batched MI355X trainer.

  * `Trainer`         the reference's single-replica loop, semantics unchanged
                      (quirks Q1-Q5 of SURVEY.md 3.2, global np.random action draws,
                      CACC test episode after every training episode,
                      train_reward.csv).  It drives any env with the reference
                      duck-type (envs.cacc_env.CACCEnv is the GPU E=1 adapter).
  * `BatchedTrainer`  E lock-stepped replicas, everything resident in HBM: the
                      n_step rollout (policy step x2, action draw, env step) is
                      one captured hipGraph, the A2C update runs once per batch,
                      data-parallel ranks exchange one flat gradient all-reduce.
  * `Evaluator`       the reference's evaluation loop over test seeds.
"""
import glob
import logging
import os
import shutil
import time

import numpy as np
import pandas as pd
import torch

from . import ops


# ------------------------------------------------------------------ small helpers (the names main.py imports, utils.py:11-60)
check_dir = os.path.exists


def copy_file(src, dst_dir):
    shutil.copy(src, dst_dir)


def find_file(cur_dir, suffix='.ini'):
    hits = sorted(glob.glob(os.path.join(cur_dir, '*' + suffix)))
    if not hits:
        logging.error('Cannot find %s file' % suffix)
        return None
    return hits[0]


def init_dir(base_dir, pathes=('log', 'data', 'model')):
    """exist_ok everywhere: under torchrun every rank calls this on the same fresh base dir."""
    dirs = {}
    for path in pathes:
        dirs[path] = os.path.join(base_dir, path) + '/'
        os.makedirs(dirs[path], exist_ok=True)
    return dirs


def init_log(log_dir, rank=0):
    """One log file per rank (`<time>.log`, `<time>.rank<r>.log` for r > 0); only rank 0 echoes to the console."""
    name = '%s/%d%s.log' % (log_dir, time.time(), '' if rank == 0 else '.rank%d' % rank)
    handlers = [logging.FileHandler(name)] + ([logging.StreamHandler()] if rank == 0 else [])
    logging.basicConfig(format='%(asctime)s [%(levelname)s] %(message)s', level=logging.INFO, handlers=handlers)


class Counter:
    """The reference's step counter (utils.py:70-97: same attributes and methods, main.py / Trainer drive it) over a plain
    integer; `advance(n)` is the batched path's n environment steps at once."""

    def __init__(self, total_step, test_step, log_step):
        self.total_step, self.test_step, self.log_step = total_step, test_step, log_step
        self.cur_step = self.cur_test_step = 0
        self.stop = False

    def advance(self, n):
        self.cur_step += n
        return self.cur_step

    def next(self):
        return self.advance(1)

    def should_test(self):
        due = self.cur_step - self.cur_test_step >= self.test_step
        if due:
            self.cur_test_step = self.cur_step
        return due

    def should_log(self):
        return self.cur_step % self.log_step == 0

    def should_stop(self):
        return self.stop or self.cur_step >= self.total_step


class SummaryWriter:
    """Stand-in for tf.summary.FileWriter (tensorboard is not part of this path): scalars go to
    a JSON-lines file `<log_dir>/scalars.jsonl`."""

    def __init__(self, log_dir=None):
        self.path = None if log_dir is None else os.path.join(log_dir, 'scalars.jsonl')
        self._rows = []

    def add_scalar(self, tag, value, global_step):
        self._rows.append('{"tag": "%s", "value": %.9g, "step": %d}' % (tag, float(value), int(global_step)))

    def flush(self):
        if self.path and self._rows:
            with open(self.path, 'a') as f:
                f.write('\n'.join(self._rows) + '\n')
        self._rows = []


# ------------------------------------------------------------------ reference-compatible loop
class Trainer:
    """The reference's Trainer (utils.py:100-254) for ONE replica."""

    def __init__(self, env, model, global_counter, summary_writer, output_path=None):
        self.cur_step = 0
        self.global_counter = global_counter
        self.env = env
        self.agent = self.env.agent
        self.model = model
        self.n_step = self.model.n_step
        self.summary_writer = summary_writer
        assert self.env.T % self.n_step == 0
        self.data = []
        self.output_path = output_path
        self.env.train_mode = True

    def _add_summary(self, reward, global_step, is_train=True):
        if self.summary_writer is not None:
            self.summary_writer.add_scalar('train_reward' if is_train else 'test_reward', reward, global_step)

    def _get_policy(self, ob, done, mode='train'):
        if self.agent.startswith('ma2c'):
            self.ps = self.env.get_fingerprint()
            policy = self.model.forward(ob, done, self.ps)
        else:
            policy = self.model.forward(ob, done)
        action = []
        for pi in policy:
            if mode == 'train':
                action.append(np.random.choice(np.arange(len(pi)), p=pi))     # global MT19937 stream (Q5)
            else:
                action.append(np.argmax(pi))
        return policy, np.array(action)

    def _get_value(self, ob, done, action):
        if self.agent.startswith('ma2c'):
            value = self.model.forward(ob, done, self.ps, np.array(action), 'v')
        else:
            self.naction = self.env.get_neighbor_action(action)
            value = self.model.forward(ob, done, self.naction, 'v')
        return value

    def _log_episode(self, global_step, mean_reward, std_reward):
        self.data.append({'agent': self.agent, 'step': global_step, 'test_id': -1,
                          'avg_reward': mean_reward, 'std_reward': std_reward})
        self._add_summary(mean_reward, global_step)
        if self.summary_writer is not None:
            self.summary_writer.flush()

    def explore(self, prev_ob, prev_done):
        ob, done = prev_ob, prev_done
        for _ in range(self.n_step):
            policy, action = self._get_policy(ob, done)          # pre-decision
            value = self._get_value(ob, done, action)            # post-decision (double-stepped LSTM: Q1)
            self.env.update_fingerprint(policy)
            next_ob, reward, done, global_reward = self.env.step(action)
            self.episode_rewards.append(global_reward)
            global_step = self.global_counter.next()
            self.cur_step += 1
            if self.agent.startswith('ma2c'):
                self.model.add_transition(ob, self.ps, action, reward, value, done)
            else:
                self.model.add_transition(ob, self.naction, action, reward, value, done)
            if self.global_counter.should_log():
                logging.info('Training: global step %d, episode step %d, a: %s, r: %.2f, train r: %.2f, done: %r' %
                             (global_step, self.cur_step, str(action), global_reward, np.mean(reward), done))
            if done:                                             # terminal check inside the batch loop
                break
            ob = next_ob
        if done:
            R = np.zeros(self.model.n_agent)
        else:
            _, action = self._get_policy(ob, done)               # advances the LSTM state once more (Q2)
            R = self._get_value(ob, done, action)
        return ob, done, R

    def perform(self, test_ind, gui=False):
        ob = self.env.reset(gui=gui, test_ind=test_ind)
        rewards = []
        done = True                                              # pre-decision done resets the LSTM (Q3)
        self.model.reset()
        while True:
            if self.env.name.startswith('atsc'):
                policy, action = self._get_policy(ob, done)
            else:
                policy, action = self._get_policy(ob, done, mode='test')   # CACC: deterministic test policy
            self.env.update_fingerprint(policy)
            next_ob, reward, done, global_reward = self.env.step(action)
            rewards.append(global_reward)
            if done:
                break
            ob = next_ob
        return np.mean(np.array(rewards)), np.std(np.array(rewards))

    def run(self):
        while not self.global_counter.should_stop():
            ob = self.env.reset()
            done = True
            self.model.reset()
            self.cur_step = 0
            self.episode_rewards = []
            while True:
                ob, done, R = self.explore(ob, done)
                dt = self.env.T - self.cur_step
                global_step = self.global_counter.cur_step
                self.model.backward(R, dt, self.summary_writer, global_step)
                if done:
                    self.env.terminate()
                    break
            rewards = np.array(self.episode_rewards)
            mean_reward, std_reward = np.mean(rewards), np.std(rewards)
            if not self.env.name.startswith('atsc'):
                # CACC: the logged reward is that of a deterministic TEST episode (utils.py:246-251)
                self.env.train_mode = False
                mean_reward, std_reward = self.perform(-1)
                self.env.train_mode = True
            self._log_episode(global_step, mean_reward, std_reward)
        if self.output_path is not None:
            pd.DataFrame(self.data).to_csv(self.output_path + 'train_reward.csv')


class Evaluator(Trainer):
    """utils.py:311-336: one `perform` per test seed, then env.output_data()."""

    def __init__(self, env, model, output_path, gui=False):
        self.env = env
        self.model = model
        self.agent = self.env.agent
        self.env.train_mode = False
        self.test_num = self.env.test_num
        self.output_path = output_path
        self.gui = gui

    def run(self):
        is_record = not self.gui
        self.env.cur_episode = 0
        self.env.init_data(is_record, False, self.output_path)
        rewards = []
        for test_ind in range(self.test_num):
            reward, _ = self.perform(test_ind, gui=self.gui)
            self.env.terminate()
            logging.info('test %i, avg reward %.2f' % (test_ind, reward))
            rewards.append(reward)
            self.env.collect_tripinfo()
        self.env.output_data()
        return rewards


# ------------------------------------------------------------------ the MI355X loop
class BatchedTrainer:
    """E lock-stepped replicas on one GPU (one process per GPU under torch.distributed).

    One `run_batch()` = the reference's `explore` + `model.backward` for every replica:
    n_step lock-steps (each: policy step, Philox action draw, value re-step, env kernel,
    transition store), the bootstrap value, then one A2C update.  Episodes end only at batch
    boundaries (Q4) so every replica contributes exactly n_step transitions; finished
    replicas are re-initialised by the env kernel's fused auto-reset at the last step and
    their LSTM state / fingerprints are cleared after the update, which is what the
    reference does at the next `env.reset(); model.reset()`.
    """

    def __init__(self, env, model, global_counter=None, summary_writer=None, output_path=None,
                 use_graph=True, rank=0, world_size=1, save_activations=True, compact_obs=True, fused_encode=True):
        self.env, self.model = env, model
        # uncoupled nets: the rollout's policy steps double as the forward pass of the update (models.py)
        self.saved_acts = bool(save_activations) and model.enable_saved_activations()
        # CACC: compact observations (own features only; the encoder gathers the neighbours) -- SURVEY.md 8d's layout
        self.compact_obs = bool(compact_obs) and hasattr(env, 'set_compact_obs') and model.enable_compact_obs() and \
            env.set_compact_obs(True)
        # CACC: the env kernel runs the next lock-step's input encoders behind its step (csrc/cacc.hip cacc_step_encode_kernel)
        self.fused_encode = bool(fused_encode) and self.saved_acts and self.compact_obs and \
            getattr(env, 'supports_fused_encode', False) and env.device.type == 'cuda' and \
            model.policy.fused_env_encode(model.buf_fp[1], model.encode_target(1)) is not None
        self.E, self.N = env.E, env.n_agent
        self.n_step = model.n_step
        assert env.T % self.n_step == 0
        assert getattr(env, 'batch_size', None) in (None, self.n_step), \
            'ENV_CONFIG batch_size must equal MODEL_CONFIG batch_size (episodes end at batch boundaries)'
        self.global_counter = global_counter
        self.summary_writer = summary_writer
        self.output_path = output_path
        self.rank, self.world_size = rank, world_size
        d = env.device
        self.device = d
        self.action_boot = torch.zeros(self.E, self.N, dtype=torch.uint8, device=d)
        self.done_pre = model.buf_done_pre[0]                                    # [E] f32, episode start (Q3)
        self.done_pre.fill_(1.0)
        self.zero_done = torch.zeros(self.E, dtype=torch.float32, device=d)
        self.step_dev = torch.zeros((), dtype=torch.int64, device=d)            # global lock-step (Philox counter)
        self.R_end = torch.zeros(self.N, self.E, dtype=torch.float32, device=d)
        self.buf_g = torch.zeros(self.n_step, self.E, dtype=torch.float32, device=d)
        self.buf_rraw = torch.zeros_like(model.buf_r)                            # raw rewards written by the env kernel
        self.last_done = model.buf_done_post[self.n_step - 1]                    # view: done flags of the last lock-step
        # per-replica running episode statistics (sum, sum of squares, length) of the global reward
        self.ep_sum = torch.zeros(self.E, dtype=torch.float64, device=d)
        self.ep_sq = torch.zeros(self.E, dtype=torch.float64, device=d)
        self.ep_len = torch.zeros(self.E, dtype=torch.float64, device=d)
        self.fin = torch.zeros(4, dtype=torch.float64, device=d)   # episodes, sum(mean), sum(std), collisions
        self.use_graph = use_graph and d.type == 'cuda'
        self.graph = None
        # coupled nets on the in-launch hand-off kernels (one-launch lock-step / BPTT): every batch is checked and, if a wave
        # timed out, re-run on the launch-per-step kernels from the state it started from (see run_batch)
        self.handoff_guard = self.saved_acts and model.policy.coupled and d.type == 'cuda' and ops.handoff_enabled()
        self.handoff_fallbacks = 0
        self._shadow = [torch.empty_like(t) for t in env.state_tensors() + [self.step_dev]] if self.handoff_guard else None
        self.data = []
        self.n_batches = 0
        env.train_mode = True
        model.reset_states()
        model.masked_steps = (0,)             # only the first lock-step of a batch can start an episode (Q4)
        model.t = 0
        model.buf_x[0].copy_(env.reset())

    # -- the n_step rollout (graph body): every kernel reads / writes rollout-buffer slots directly
    def _rollout(self):
        env, model = self.env, self.model
        T = self.n_step
        model.t = 0
        if self.handoff_guard:                # what the rollout mutates and a re-run must start from (a few small copies)
            for s_, t_ in zip(self._shadow, env.state_tensors() + [self.step_dev]):
                s_.copy_(t_)
        # Philox step = batch base (device counter, advanced once per batch) + slot offset baked into the graph
        fused = self.fused_encode
        for t in range(T):
            action = model.act(self.done_pre if t == 0 else self.zero_done, mode=ops.SAMPLE_PHILOX, seed=env.seed,
                               env_id_base=env.env_id_base, step=t, step_dev=self.step_dev, done_is_zero=(t > 0),
                               pre_encoded=fused and t > 0)
            # fused: the env kernel also runs lock-step t + 1's input encoders on the observation it just produced and on the
            # policies the step above wrote (the next fingerprints) -- one launch instead of two per lock-step
            enc = model.policy.fused_env_encode(model.buf_fp[t + 1], model.encode_target(t + 1)) if fused else None
            env.step(action, auto_reset=(t == T - 1), obs_out=model.buf_x[t + 1], reward_out=self.buf_rraw[t],
                     done_out=model.buf_done_post[t], greward_out=self.buf_g[t], **({'encode': enc} if fused else {}))
            model.t = t + 1
        # bootstrap value for unfinished replicas (utils.py:192-196); finished ones get R = 0
        v = model.bootstrap(self.zero_done, self.action_boot, mode=ops.SAMPLE_PHILOX, seed=env.seed,
                            env_id_base=env.env_id_base, step=T, step_dev=self.step_dev, done_is_zero=True, pre_encoded=fused)
        self.step_dev.add_(T + 1)
        self.R_end.copy_(v * (1.0 - self.last_done.to(torch.float32)).view(1, -1))

    def rollout(self):
        if not self.use_graph:
            self._rollout()
            return
        if self.graph is None:
            # warm-up on a side stream (allocator + rocBLAS handles), restoring all mutated state after
            snap = self._snapshot()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._rollout()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._restore(snap)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._rollout()
            self._restore(snap)
            # host-side per-batch state the captured rollout set (a replay runs no Python): the update's "the rollout saved the
            # encoder outputs" flag is cleared by every update and must be raised again after every replay
            self._graph_flags = {k: bool(getattr(self.model.policy, k, False)) for k in ('_enc_was_saved', '_mm_was_saved')}
        self.model.t = 0
        self.graph.replay()
        self.model.t = self.n_step
        for k, v in self._graph_flags.items():
            setattr(self.model.policy, k, v)

    def _state_tensors(self):
        m = self.model
        return self.env.state_tensors() + [m.h_fw, m.c_fw, m.buf_x[0], m.buf_fp[0], self.step_dev, self.done_pre]

    def _snapshot(self):
        return [t.clone() for t in self._state_tensors()]

    def _restore(self, snap):
        for t, s in zip(self._state_tensors(), snap):
            t.copy_(s)

    def run_batch(self):
        """One rollout + update.  Returns nothing; statistics stay on the device until `stats()`."""
        self.rollout()
        m = self.model
        m.load_rewards(self.buf_rraw)
        m.update(self.R_end, rotate=False)
        if self.handoff_guard and ops.handoff_poisoned(self.device):
            self._recover_from_handoff_timeout()
        # episode statistics, then the hand-over to the next batch in one call: finished replicas start a new episode (the
        # env already auto-reset them) with zero recurrent state and uniform fingerprints -- what the reference does at its
        # next `env.reset(); model.reset()` --, states_bw <- states_fw, slot T of the rollout buffers -> slot 0
        T = self.n_step
        ops.batch_epilogue(self.buf_g, self.last_done, self.ep_sum, self.ep_sq, self.ep_len, self.fin, self.env.T,
                           m.h_fw, m.c_fw, m.h_bw, m.c_bw, m.buf_fp[T], m.buf_fp[0], m.fp_uniform, m.buf_x[T], m.buf_x[0],
                           self.done_pre)
        self.n_batches += 1
        if self.global_counter is not None:
            # the counter (like the reference's global step and the lr schedule) counts LOCK-steps, i.e. environment
            # steps per replica: `total_step` of the ini keeps its meaning (1e6 -> 16 667 updates at n_step 60)
            self.global_counter.advance(self.n_step)

    def _recover_from_handoff_timeout(self):
        """A wave of an in-launch hand-off kernel gave up waiting during this batch (its neighbour block was not resident: the
        device is shared, masked or profiled).  The optimiser step refused the batch on the device (nothing was applied); here
        the batch is rewound to the state it started from, the process is pinned to the launch-per-step kernels, and the
        batch is run again -- the weights end up exactly where a run without the one-launch kernels puts them."""
        m, dev = self.model, self.device
        logging.warning('in-launch hand-off timed out (batch %d): re-running the batch on the launch-per-step kernels and '
                        'keeping them for the rest of the run' % self.n_batches)
        ops.disable_inkernel_handoff()
        ops.handoff_clear(dev)
        for s_, t_ in zip(self._shadow, self.env.state_tensors() + [self.step_dev]):
            t_.copy_(s_)
        m.h_fw.copy_(m.H_all[:, 0])            # the persistent state the rollout started from (its bootstrap step overwrote it)
        m.c_fw.copy_(m.C_all[:, 0])
        m.lr_scheduler.n -= self.n_step        # update() advanced the schedule
        self.graph = None                      # re-capture: the rollout now takes the two-launch lock-step
        self.handoff_guard = False
        self.rollout()
        m.load_rewards(self.buf_rraw)
        m.update(self.R_end, rotate=False)
        ops.check_coupled_status(dev)
        self.handoff_fallbacks += 1

    def stats(self, reset=True):
        """(episodes finished, mean of episode-mean reward, mean of episode-std, collisions) since last call."""
        f = self.fin.cpu().numpy().copy()
        ops.check_coupled_status()            # (the copy above synchronised) a wave of the coupled BPTT gave up waiting?
        if reset:
            self.fin.zero_()
        n = max(f[0], 1.0)
        return dict(episodes=int(f[0]), avg_reward=f[1] / n, std_reward=f[2] / n, collisions=int(f[3]))

    def evaluate(self, n_envs=64, seed=None):
        """Deterministic (argmax) test episodes, the batched analogue of `perform(-1)` after a CACC
        training episode (utils.py:199-223, 246-251): train_mode False -> no soft-collision term.  The evaluation env, its state
        tensors and (with use_graph) the whole T-step episode as ONE captured hipGraph are built once per (n_envs, seed) and
        replayed: every evaluation starts from the same initial conditions (the reference re-seeds its in-training test with
        seed - 1 every time, cacc_env.py:170-175) and costs a graph replay instead of ~6 T eager launches and a new env."""
        key = (int(n_envs), self.env.seed - 1 if seed is None else int(seed))
        cache = self.__dict__.setdefault('_eval_cache', {})
        if key not in cache:
            cache[key] = self._build_eval(*key)
        ev = cache[key]
        E0 = self.model.E
        if ev['graph'] is not None:
            ev['graph'].replay()
        else:
            ev['episode']()
        assert self.model.E == E0                     # evaluation used its own state tensors only
        hist, total, steps = ev['hist'], ev['total'], ev['steps']
        self.last_eval_action_share = (hist / hist.sum().clamp_min(1)).cpu().numpy().round(4).tolist()
        per_ep = (total / steps.clamp_min(1)).cpu().numpy()
        return float(per_ep.mean()), float(per_ep.std()), int((steps < ev['env'].T).sum().item())

    def _build_eval(self, n_envs, seed):
        """The test episode as ~4 launches per lock-step: encoder, ONE fused kernel for LSTM step + actor head + arg max (the
        rollout's policy-step kernel in SAMPLE_ARGMAX mode; its policy output IS the next fingerprint), env step writing reward /
        done straight into per-step slots.  The episode statistics (alive mask, sums, action histogram) are formed from those
        slots after the last step instead of ~10 elementwise launches per step."""
        from .envs import make_batch_env
        dev, model, N = self.device, self.model, self.N
        env = make_batch_env(self.env.config, num_envs=n_envs, device=dev, seed=seed, env_id_base=10 ** 9)
        env.train_mode = False
        T, A, p = env.T, model.n_a, model.policy
        f64 = dict(dtype=torch.float64, device=dev)
        hs = [torch.zeros(N, n_envs, model.n_lstm, device=dev) for _ in range(2)]      # ping-pong: a coupled net's message term
        cs = [torch.zeros(N, n_envs, model.n_lstm, device=dev) for _ in range(2)]      # reads the others' h while h' is written
        fps = [model.fp_uniform.expand(N, n_envs, A).clone() for _ in range(2)]
        done1, done0 = torch.ones(n_envs, device=dev), torch.zeros(n_envs, device=dev)
        acts = torch.zeros(T, n_envs, N, dtype=torch.uint8, device=dev)
        G = torch.zeros(T, n_envs, device=dev)
        D = torch.zeros(T, n_envs, dtype=torch.uint8, device=dev)
        rew = torch.zeros_like(env.reward)
        total, steps, hist = torch.zeros(n_envs, **f64), torch.zeros(n_envs, **f64), torch.zeros(A, **f64)
        a_ids = torch.arange(A, device=dev).view(1, 1, 1, -1)
        fused = p.fused_heads

        def episode():
            env.episode.zero_()                       # the same test episode every time
            env.reset()
            hs[0].zero_()
            cs[0].zero_()
            fps[0].copy_(model.fp_uniform.expand_as(fps[0]))
            p.refresh_wimage()
            for t in range(T):
                a, b = t & 1, (t + 1) & 1
                enc = p.encode(env.obs, fps[a])
                if fused:
                    p.step_policy(enc, hs[a], cs[a], done1 if t == 0 else done0, hs[b], cs[b], fps[b], acts[t], t > 0,
                                  mode=ops.SAMPLE_ARGMAX)
                else:
                    p.step(enc, hs[a], cs[a], done1 if t == 0 else done0, hs[b], cs[b], t > 0)
                    with torch.no_grad():
                        fps[b].copy_(p.pi(hs[b]))
                    ops.sample_actions(fps[b], acts[t], ops.SAMPLE_ARGMAX)
                env.step(acts[t], reward_out=rew, done_out=D[t], greward_out=G[t])
            # alive[t] = no `done` before step t; an episode's statistics stop with its first done
            dd = D.to(torch.float64)
            alive = torch.cat([torch.ones(1, n_envs, **f64), torch.cumprod(1.0 - dd, dim=0)[:-1]], dim=0)
            total.copy_((G.double() * alive).sum(dim=0))
            steps.copy_(alive.sum(dim=0))
            hist.copy_(((acts.unsqueeze(-1) == a_ids).to(torch.float64) * alive.view(T, n_envs, 1, 1)).sum(dim=(0, 1, 2)))

        graph = None
        if self.use_graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                episode()                             # warm-up (allocator, library handles)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                episode()
        return dict(env=env, episode=episode, graph=graph, hist=hist, total=total, steps=steps)

    def run(self, log_every=10, eval_every=None):
        """Train until the counter says stop (`total_step` lock-steps per replica); one row per `log_every` batches.
        Row: `avg_reward` / `std_reward` = the deterministic TEST episodes (argmax policy, raw reward) for CACC, like
        the reference's train_reward.csv (utils.py:246-251), evaluated every `eval_every` rows -- default: every env.T / n_step
        rows (the reference tests once per training episode of ONE replica, utils.py:246-251; here a row already spans log_every
        batches of E replicas) and at the last row; never for ATSC, whose logged
        reward is the training episode's (utils.py:243-245) -- and carried forward on the rows in between (NaN before the first
        evaluation; `evaluated` marks the rows that ran one); `train_avg_reward` etc. = statistics of the training episodes
        finished since the last row (stochastic policy, training-mode reward)."""
        t0 = time.time()
        if eval_every is None:      # one test per env.T / n_step rows (10 for CACC: every 100 batches at the default log_every)
            eval_every = 0 if self.env.name.startswith('atsc') else max(1, self.env.T // self.n_step)
        total = self.global_counter.total_step
        if total < log_every * self.n_step:
            logging.warning('total_step %d < log_every x n_step = %d lock-steps: only the final row will be logged'
                            % (total, log_every * self.n_step))
        rows_done = 0
        last_eval = [float('nan'), float('nan'), 0]

        def log_row(final=False):
            st = self.stats()
            step = self.global_counter.cur_step
            row = {'agent': self.env.agent, 'step': step, 'test_id': -1, 'avg_reward': st['avg_reward'],
                   'std_reward': st['std_reward'], 'train_avg_reward': st['avg_reward'],
                   'train_std_reward': st['std_reward'], 'episodes': st['episodes'], 'collisions': st['collisions'],
                   'env_steps': step * self.E * self.world_size * self.N, 'wall_s': time.time() - t0}
            if eval_every:
                ran = final or rows_done % eval_every == eval_every - 1
                if ran:
                    last_eval[:] = self.evaluate()
                row.update(avg_reward=last_eval[0], std_reward=last_eval[1], test_collisions=last_eval[2], evaluated=int(ran))
            self.data.append(row)
            if self.rank == 0:
                logging.info('Training: lock-step %d, batches %d, %.0f env-steps/s, episodes %d, train r %.2f, '
                             'logged r %.2f, collisions %d'
                             % (step, self.n_batches, row['env_steps'] / max(time.time() - t0, 1e-9), st['episodes'],
                                st['avg_reward'], row['avg_reward'], st['collisions']))
                if self.summary_writer is not None:
                    self.summary_writer.add_scalar('train_reward', row['avg_reward'], step)
                    self.summary_writer.flush()

        while not self.global_counter.should_stop():
            self.run_batch()
            if self.n_batches % log_every == 0:
                log_row()
                rows_done += 1
        if self.n_batches % log_every != 0 or (eval_every and self.data and not self.data[-1].get('evaluated')):
            log_row(final=True)                           # final row: never leave train_reward.csv empty / without a test
        if self.output_path is not None and self.rank == 0:
            pd.DataFrame(self.data).to_csv(self.output_path + 'train_reward.csv')
