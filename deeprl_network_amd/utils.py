"""Rollout / training loops -- the reference's root utils.py (Counter 70-97,
Trainer 100-254, Evaluator 311-336) plus the batched MI355X trainer.

  * `Trainer`         the reference Trainer's public surface for ONE replica (quirks Q1-Q5
                      of SURVEY.md 3.2, global np.random action draws, CACC test episode
                      after every training episode, train_reward.csv), built on a
                      lock-step primitive and an episode cursor of this repo.
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


# ------------------------------------------------------------------ the E = 1 driver behind the reference's Trainer surface
class _Decision:
    """What all agents decided at one lock-step: the policies pi, the drawn (or arg-max) actions, the critic values and the
    `side` input of the critic / the transition record -- the neighbours' fingerprints for the MA2C family (stacked arrays),
    the neighbours' actions for the IA2C family (per-agent lists)."""
    __slots__ = ('policy', 'action', 'value', 'side')

    def __init__(self, policy, action, value=None, side=None):
        self.policy, self.action, self.value, self.side = policy, action, value, side


class _Cursor:
    """Where a running episode stands: the observation the next decision reads and the done flag BEFORE that decision
    (True right after a reset: it clears the recurrent state, quirk Q3)."""
    __slots__ = ('ob', 'done')

    def __init__(self, ob, done=True):
        self.ob, self.done = ob, done


def _pick(pi, greedy):
    # one uniform of the global MT19937 stream per stochastic draw (quirk Q5: the reference's np.random.choice contract)
    return int(np.argmax(pi)) if greedy else int(np.random.choice(len(pi), p=pi))


class Trainer:
    """Single-replica training loop with the reference Trainer's public surface (utils.py:100-254: constructor, `explore`,
    `perform`, `run`, `data`, `episode_rewards`, the train_reward.csv rows) and its observable call order on env and model
    (quirks Q1-Q5 of SURVEY.md 3.2, pinned by the E2E goldens).  The loop itself is organised around two pieces of this
    repo: `_lock_step` (ONE decision of all agents: policy forward, action draw, then the value forward that re-steps the
    LSTM from the state the policy forward just wrote -- Q1) and `_walk` (a generator that advances an episode cursor one
    env step at a time); `explore`, the bootstrap and `perform` are thin consumers of them.  It drives any env with the
    reference duck-type (envs.cacc_env.CACCEnv is the GPU E = 1 adapter); models without a policy (`agent == 'greedy'`,
    large_grid_env.py:30-45) decide through `model.forward(ob)` alone."""

    def __init__(self, env, model, global_counter, summary_writer, output_path=None):
        assert env.T % model.n_step == 0
        self.env, self.model = env, model
        self.agent = env.agent
        self.n_step = model.n_step
        self.global_counter, self.summary_writer, self.output_path = global_counter, summary_writer, output_path
        self.data, self.episode_rewards, self.cur_step = [], [], 0
        env.train_mode = True

    # -- agent family: MA2C models take / return stacked arrays and read the fingerprints; IA2C models take per-agent lists
    @property
    def _stacked(self):
        return self.agent.startswith('ma2c')

    def _lock_step(self, ob, done, greedy=False, with_value=True):
        """-> _Decision.  Call order on the model: forward(.., 'p') then forward(.., 'v') (the second advances nothing the
        first did not: it re-steps a scratch copy, Q1)."""
        env, model = self.env, self.model
        if self.agent == 'greedy':                               # rule-based controller: actions straight from the observation
            return _Decision(None, np.asarray(model.forward(ob)))
        fps = env.get_fingerprint() if self._stacked else None
        pi = model.forward(ob, done, fps) if self._stacked else model.forward(ob, done)
        action = np.array([_pick(p, greedy) for p in pi])
        d = _Decision(pi, action, side=fps)
        if with_value:
            if self._stacked:
                d.value = model.forward(ob, done, fps, action, 'v')
            else:
                d.side = env.get_neighbor_action(action)
                d.value = model.forward(ob, done, d.side, 'v')
        return d

    def _walk(self, cur, greedy=False, learn=False):
        """Generator over the env steps of the episode `cur` points into; the cursor is advanced BEFORE each yield (a
        terminal step leaves `cur.ob` at the last pre-step observation).  learn: every step is a transition of the model's
        n-step buffer.  Yields (decision, reward, global_reward)."""
        env = self.env
        while True:
            ob, d = cur.ob, self._lock_step(cur.ob, cur.done, greedy=greedy, with_value=learn)
            if d.policy is not None:
                env.update_fingerprint(d.policy)
            nxt, reward, cur.done, g = env.step(d.action)
            if learn:
                self.model.add_transition(ob, d.side, d.action, reward, d.value, cur.done)
            if not cur.done:
                cur.ob = nxt
            yield d, reward, g
            if cur.done:
                return

    def explore(self, prev_ob, prev_done):
        """One n_step batch from (prev_ob, prev_done) -> (ob, done, R): R = 0 after a terminal step, else the bootstrap
        value of a further decision, whose policy forward advances the recurrent state once more (Q2)."""
        cur = _Cursor(prev_ob, prev_done)
        steps = self._walk(cur, learn=True)
        for _ in range(self.n_step):                             # the terminal check lives inside the batch (Q4)
            step = next(steps, None)
            if step is None:
                break
            d, reward, g = step
            self.episode_rewards.append(g)
            self.cur_step += 1
            now = self.global_counter.next()
            if self.global_counter.should_log():
                logging.info('Training: global step %d, episode step %d, a: %s, r: %.2f, train r: %.2f, done: %r'
                             % (now, self.cur_step, d.action, g, np.mean(reward), cur.done))
        R = np.zeros(self.model.n_agent) if cur.done else self._lock_step(cur.ob, cur.done).value
        return cur.ob, cur.done, R

    def perform(self, test_ind, gui=False):
        """One evaluation episode -> (mean, std) of its global rewards.  CACC is safety critical: arg-max policy; ATSC keeps
        the stochastic on-policy one (utils.py:199-223)."""
        cur = _Cursor(self.env.reset(gui=gui, test_ind=test_ind))
        self.model.reset()
        g = np.array([g for _, _, g in self._walk(cur, greedy=not self.env.name.startswith('atsc'))])
        return g.mean(), g.std()

    def _train_episode(self):
        """env.reset -> model.reset -> (explore, backward) until the terminal batch -> (mean, std, global step)."""
        cur = _Cursor(self.env.reset())
        self.model.reset()
        self.cur_step, self.episode_rewards = 0, []
        while True:
            cur.ob, cur.done, R = self.explore(cur.ob, cur.done)
            at = self.global_counter.cur_step
            self.model.backward(R, self.env.T - self.cur_step, self.summary_writer, at)
            if cur.done:
                break
        self.env.terminate()
        g = np.array(self.episode_rewards)
        return g.mean(), g.std(), at

    def _record(self, at, mean, std):
        self.data.append(dict(agent=self.agent, step=at, test_id=-1, avg_reward=mean, std_reward=std))
        if self.summary_writer is not None:
            self.summary_writer.add_scalar('train_reward', mean, at)
            self.summary_writer.flush()

    def run(self):
        while not self.global_counter.should_stop():
            mean, std, at = self._train_episode()
            if not self.env.name.startswith('atsc'):
                # CACC logs a deterministic TEST episode instead (other reward terms, other policy: utils.py:246-251)
                self.env.train_mode = False
                try:
                    mean, std = self.perform(-1)
                finally:
                    self.env.train_mode = True
            self._record(at, mean, std)
        if self.output_path is not None:
            pd.DataFrame(self.data).to_csv(self.output_path + 'train_reward.csv')


class Evaluator(Trainer):
    """`perform` over the env's test seeds, then the env's recorded CSVs (utils.py:311-336)."""

    def __init__(self, env, model, output_path, gui=False):
        self.env, self.model, self.agent = env, model, env.agent
        self.output_path, self.gui = output_path, gui
        self.test_num = env.test_num
        env.train_mode = False

    def run(self):
        env = self.env
        env.cur_episode = 0
        env.init_data(not self.gui, False, self.output_path)
        means = []
        for k in range(self.test_num):
            means.append(self.perform(k, gui=self.gui)[0])
            env.terminate()
            logging.info('test %i, avg reward %.2f' % (k, means[-1]))
            env.collect_tripinfo()
        env.output_data()
        return means


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
                 use_graph=True, rank=0, world_size=1, save_activations=True, compact_obs=True, fused_encode=True,
                 capture_update=True, rearm_after=200, keep_graphs=False):
        self.env, self.model = env, model
        # keep_graphs: the captured hipGraph_t objects stay inspectable (torch.cuda.CUDAGraph.raw_cuda_graph; tools/graph_nodes.py)
        self.keep_graphs = bool(keep_graphs) or os.environ.get('NMARL_KEEP_GRAPHS', '0') == '1'
        # uncoupled nets: the rollout's policy steps double as the forward pass of the update (models.py)
        self.saved_acts = bool(save_activations) and model.enable_saved_activations()
        # CACC: compact observations (own features only; the encoder gathers the neighbours) -- SURVEY.md 8d's layout
        self.compact_obs = bool(compact_obs) and hasattr(env, 'set_compact_obs') and model.enable_compact_obs() and \
            env.set_compact_obs(True)
        self._want_fused_encode = bool(fused_encode)
        self._select_lock_step_form()
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
        # the A2C update (and, where no host decision sits in between, the batch epilogue) as hipGraphs too: the update's 44-71
        # launches were issued eagerly, with 0.36-0.66 ms of idle device per batch between them; captured
        # after the first (eager) batch has tuned / warmed the library GEMMs.  NMARL_CAPTURE_UPDATE=0 keeps it eager
        self.capture_update = bool(capture_update) and self.use_graph and os.environ.get('NMARL_CAPTURE_UPDATE', '1') != '0'
        self._upd = None                      # dict(grads=CUDAGraph, apply=CUDAGraph or None, epilogue_inside=bool)
        self.update_capture_error = None
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=d)
        self._lr_dev_host = None
        self._keepalive = []                  # tensors whose addresses captured graphs hold
        # after a hand-off time-out the launch-per-step forms are used; after `rearm_after` clean batches the one-launch forms
        # are tried again (a transient -- a profiler helper, a second process that has left -- no longer costs the whole run);
        # every further time-out doubles the wait
        self.rearm_after = int(rearm_after)
        self._rearm_wait, self._clean_since_fallback = int(rearm_after), 0
        # coupled nets on the in-launch hand-off kernels (one-launch lock-step / BPTT).  A wave that gives up waiting raises the
        # device's status word; while it is raised NOTHING of a batch is committed on the device -- the optimiser step refuses,
        # the batch epilogue (statistics + hand-over) refuses, the next rollout does not overwrite the start-of-batch snapshot --
        # so the host need not look at the word before it launches the next batch: it reads it one batch LATE through a pinned
        # non-blocking copy (no host synchronisation on the critical path) and then re-runs the refused batches on the
        # launch-per-step kernels from the snapshot (`_recover_from_handoff_timeout`)
        self._wants_guard = self.saved_acts and bool(model.policy.coupled) and d.type == 'cuda'
        self.handoff_guard = self._wants_guard and ops.handoff_enabled()
        self.handoff_fallbacks = 0
        self._shadow = [torch.empty_like(t) for t in self._shadow_tensors()] if self._wants_guard else None
        self._status = ops.handoff_status(d) if self._wants_guard else None
        self._probes = [(torch.zeros(4, dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(2)] if self._wants_guard else None
        self._pending = None                  # (pinned copy of the status words, event) of the batch launched last
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
        if self.handoff_guard:                # what the rollout mutates and a re-run must start from: ONE launch, and none of it
            ops.copy_multi(zip(self._shadow, self._shadow_tensors()), skip_if=self._status[:1])    # once the status word is raised
        # Philox step = batch base (device counter, advanced once per batch) + slot offset baked into the graph
        fused = self.fused_encode
        for t in range(T):
            outs = dict(obs_out=model.buf_x[t + 1], reward_out=self.buf_rraw[t], done_out=model.buf_done_post[t], greward_out=self.buf_g[t])
            if self.env_in_kernel:            # the whole lock-step -- encoders, policy step, draw, value re-step, env step -- in ONE launch
                model.act(self.done_pre if t == 0 else self.zero_done, mode=ops.SAMPLE_PHILOX, seed=env.seed,
                          env_id_base=env.env_id_base, step=t, step_dev=self.step_dev, done_is_zero=(t > 0),
                          env_step=env.inkernel_step(auto_reset=(t == T - 1), **outs))
                model.t = t + 1
                continue
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
        torch.mul(v, (1.0 - self.last_done.to(torch.float32)).view(1, -1), out=self.R_end)     # (out=: no temporary + memcpy node)

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
            self.graph = self._new_graph()
            with torch.cuda.graph(self.graph):
                self._rollout()
            self._restore(snap)
            # host-side per-batch state the captured rollout set (a replay runs no Python): the update's "the rollout saved the
            # encoder outputs" flag is cleared by every update and must be raised again after every replay
            self._graph_flags = {k: getattr(self.model.policy, k, False) for k in ('_enc_was_saved', '_mm_was_saved', '_bits_steps')}
        self.model.t = 0
        self.graph.replay()
        self.model.t = self.n_step
        for k, v in self._graph_flags.items():
            setattr(self.model.policy, k, v)

    def _new_graph(self):
        return torch.cuda.CUDAGraph(keep_graph=True) if self.keep_graphs else torch.cuda.CUDAGraph()

    def _select_lock_step_form(self):
        """How many launches a lock-step is (decided at construction and again whenever the in-launch hand-off kernels are switched
        off / on: a coupled net's one-launch form exists only with them)."""
        env, model = self.env, self.model
        if env.device.type == 'cuda' and self.saved_acts:
            model.policy.refresh_wimage()     # (a coupled net's one-launch forms exist once its message image does)
        # CACC: the env kernel runs the next lock-step's input encoders behind its step (csrc/cacc.hip cacc_step_encode_kernel)
        # ... unless the lock-step kernel runs the encoders itself (csrc/lstm_mfma.hip ENC: IA2C-FP, NeurComm), then the env step stays alone
        self.enc_in_kernel = self.saved_acts and self.compact_obs and env.device.type == 'cuda' and \
            model.policy.enc_in_kernel(env.E, True)
        # ... and the env step as well, behind the action draw of the same launch: ONE launch per lock-step (ENV block of the kernel)
        self.env_in_kernel = self.enc_in_kernel and hasattr(env, 'inkernel_step') and env.n_agent == 8 and ops.step_env_supported() and \
            model.policy.env_step_in_kernel
        # the synthetic grid under CommNet: the env step is a ROLE of the one-launch lock-step (its observation encoder is inside
        # already), run by the compute units the 25 x ceil(E / 128) LSTM blocks leave idle (csrc/lstm_mfma.hip GENV)
        if not self.env_in_kernel and self.saved_acts and self.compact_obs and env.device.type == 'cuda' and \
                getattr(env, 'inkernel_step_supported', None) is not None and model.policy.coupled and \
                getattr(model.policy, 'encodes_in_step', None) is not None and 'ENC' in model.policy._extra and \
                os.environ.get('NMARL_GRID_ENV_IN_KERNEL', '1') != '0':
            self.env_in_kernel = bool(model.policy.encodes_in_step(env.E, True) and env.inkernel_step_supported())
        self.fused_encode = self._want_fused_encode and self.saved_acts and self.compact_obs and not self.enc_in_kernel and \
            getattr(env, 'supports_fused_encode', False) and env.device.type == 'cuda' and \
            model.policy.fused_env_encode(model.buf_fp[1], model.encode_target(1)) is not None

    def _shadow_tensors(self):
        """What a rollout overwrites and the batch epilogue does not restore: the env, the Philox counter, the persistent
        recurrent state (its bootstrap step advances it, Q2)."""
        return self.env.state_tensors() + [self.step_dev, self.model.h_fw, self.model.c_fw]

    def _state_tensors(self):
        m = self.model
        return self.env.state_tensors() + [m.h_fw, m.c_fw, m.buf_x[0], m.buf_fp[0], self.step_dev, self.done_pre]

    def _snapshot(self):
        return [t.clone() for t in self._state_tensors()]

    def _restore(self, snap):
        for t, s in zip(self._state_tensors(), snap):
            t.copy_(s)

    def _epilogue(self):
        """Episode statistics, then the hand-over to the next batch in one call: finished replicas start a new episode (the
        env already auto-reset them) with zero recurrent state and uniform fingerprints -- what the reference does at its
        next `env.reset(); model.reset()` --, states_bw <- states_fw, slot T of the rollout buffers -> slot 0."""
        m, T = self.model, self.n_step
        m.policy.invalidate_cached_msg()       # (the epilogue kernel zeroes finished replicas' h through raw pointers)
        ops.batch_epilogue(self.buf_g, self.last_done, self.ep_sum, self.ep_sq, self.ep_len, self.fin, self.env.T,
                           m.h_fw, m.c_fw, m.h_bw, m.c_bw, m.buf_fp[T], m.buf_fp[0], m.fp_uniform, m.buf_x[T], m.buf_x[0],
                           self.done_pre, skip_if=self._status[:1] if self.handoff_guard else None)

    def _capture_update(self):
        """Capture the update of a batch as hipGraphs (after at least one eager batch: the library GEMMs are tuned, every
        lazily built table and workspace exists).  One graph [rewards, returns, loss, backward, clip + RMSProp, epilogue];
        with several ranks two, around the eager gradient all-reduce.  Capturing runs no kernel; the host-side state the
        captured Python code touches is put back."""
        m = self.model
        split = m.dist_group is not None
        inside = True
        tun = None
        try:
            import torch.cuda.tunable as tunable
            if tunable.is_enabled() and tunable.tuning_is_enabled():
                tun = tunable
                tunable.tuning_enable(False)          # a timing loop inside a capture would invalidate it (every shape is tuned by now)
        except Exception:
            tun = None
        host = (m.t, m.policy._enc_was_saved, m.policy._mm_was_saved, m.policy._bits_steps)
        ops.keepalive_begin(self._keepalive)
        try:
            torch.cuda.synchronize()
            g1, g2 = self._new_graph(), None
            with torch.cuda.graph(g1):
                m.load_rewards(self.buf_rraw)
                m.update_grads(self.R_end)
                if not split:
                    m.update_apply(0.0, rotate=False, lr_dev=self.lr_dev)
                    if inside:
                        self._epilogue()
            if split:
                g2 = self._new_graph()
                with torch.cuda.graph(g2):
                    m.update_apply(0.0, rotate=False, lr_dev=self.lr_dev)
                    if inside:
                        self._epilogue()
            self._upd = dict(grads=g1, apply=g2, epilogue_inside=inside)
        finally:
            ops.keepalive_end()
            m.t, m.policy._enc_was_saved, m.policy._mm_was_saved, m.policy._bits_steps = host
            if tun is not None:
                tun.tuning_enable(True)

    def _update(self):
        """The update + hand-over of the batch the rollout just produced: replayed graphs, or eager launches."""
        m = self.model
        if self.capture_update and self._upd is None and self.update_capture_error is None and self.n_batches >= 1:
            try:
                self._capture_update()
            except Exception as ex:            # keep training on the eager path, and say so (bench.py reports it)
                self.update_capture_error = repr(ex)
                self._upd = None
                logging.warning('update capture failed, staying eager: %r' % (ex,))
                torch.cuda.synchronize()
        if self._upd is None:
            m.load_rewards(self.buf_rraw)
            m.update(self.R_end, rotate=False)
            return False
        lr = m.update_begin()
        if lr != self._lr_dev_host:            # the schedule moved (never for lr_decay = constant): one tiny launch
            self.lr_dev.fill_(lr)
            self._lr_dev_host = lr
        m.t = self.n_step
        self._upd['grads'].replay()
        if self._upd['apply'] is not None:
            m.update_reduce()
            self._upd['apply'].replay()
        m.update_end()
        return self._upd['epilogue_inside']

    def run_batch(self):
        """One rollout + update.  Returns nothing; statistics stay on the device until `stats()`."""
        self.rollout()
        if not self._update():
            self._epilogue()
        self.n_batches += 1
        if self.global_counter is not None:
            # the counter (like the reference's global step and the lr schedule) counts LOCK-steps, i.e. environment
            # steps per replica: `total_step` of the ini keeps its meaning (1e6 -> 16 667 updates at n_step 60)
            self.global_counter.advance(self.n_step)
        if self.handoff_guard:
            self._probe_handoff_status()
        elif self.handoff_fallbacks and self._wants_guard:
            self._clean_since_fallback += 1
            if self.rearm_after > 0 and self._clean_since_fallback >= self._rearm_wait:
                self._rearm()

    def _probe_handoff_status(self):
        """Queue a non-blocking copy of the status words behind the batch just launched, then look at the copy queued behind the
        batch BEFORE it (long finished, or finishing while this batch runs: the host stays at most one batch ahead of the
        device and never waits for the batch it has just launched)."""
        host, ev = self._probes[self.n_batches & 1]
        host.copy_(self._status, non_blocking=True)
        ev.record()
        prev, self._pending = self._pending, (host, ev)
        if prev is not None:
            prev[1].synchronize()
            if int(prev[0][0]) != 0:
                self._recover_from_handoff_timeout(batches=2)      # the poisoned batch and the one launched behind it

    def flush(self):
        """Look at the status of the batch launched last as well (end of a run, before statistics are read)."""
        if self._pending is not None:
            prev, self._pending = self._pending, None
            prev[1].synchronize()
            if int(prev[0][0]) != 0:
                self._recover_from_handoff_timeout(batches=1)

    def _drop_graphs(self):
        """Forget the captured graphs (the kernels a lock-step / an update launches are about to change).  The device is idle
        when they go (synchronised here: a rare event), so only ONE generation is parked -- until its replacements exist."""
        torch.cuda.synchronize()
        self._parked = (self.graph, self._upd, self._keepalive)      # (replaces the generation parked before)
        self.graph, self._upd, self._keepalive = None, None, []

    def _rearm(self):
        """Try the one-launch hand-off kernels again after a run of clean batches on the launch-per-step forms."""
        logging.info('re-arming the in-launch hand-off kernels after %d clean batches' % self._clean_since_fallback)
        ops.enable_inkernel_handoff()
        self._clean_since_fallback = 0
        self._rearm_wait *= 2                  # the next time-out waits twice as long
        # the guard follows the switch alone (as in __init__): ops.bptt_coupled picks its hand-off form from it, whether or not
        # the lock-step has a one-launch form at this size
        self.handoff_guard = self._wants_guard and ops.handoff_enabled()
        self._pending = None
        self._select_lock_step_form()
        self._drop_graphs()

    def _recover_from_handoff_timeout(self, batches=1):
        """A wave of an in-launch hand-off kernel gave up waiting `batches` batches ago (its neighbour block was not resident:
        the device is shared, masked or profiled).  Nothing of these batches was committed on the device: the optimiser steps
        were refused, the epilogues were refused, the snapshot in `_shadow` still is the state the first of them started from.
        Here the host rewinds its own counters, selects the launch-per-step kernels (until `_rearm`) and runs the batches again
        -- the weights end up exactly where a run without the one-launch kernels puts them."""
        m, dev = self.model, self.device
        torch.cuda.synchronize()
        logging.warning('in-launch hand-off timed out (batch %d): re-running %d batch(es) on the launch-per-step kernels and '
                        'keeping them for the next %d batches' % (self.n_batches - batches, batches, self._rearm_wait))
        ops.disable_inkernel_handoff()
        ops.handoff_clear(dev)
        if hasattr(self.env, 'clear_inkernel_words'):
            self.env.clear_inkernel_words()
        for s_, t_ in zip(self._shadow, self._shadow_tensors()):
            t_.copy_(s_)
        m.policy.invalidate_cached_msg()
        m.lr_scheduler.rewind(self.n_step * batches)       # the refused updates advanced the schedule
        self.n_batches -= batches
        if self.global_counter is not None:
            self.global_counter.advance(-self.n_step * batches)
        self.handoff_guard, self._pending = False, None
        self._select_lock_step_form()
        self._drop_graphs()                    # re-capture: the rollout now takes the two-launch lock-step, the update the step-wise BPTT
        self._clean_since_fallback = -batches  # (the re-run batches below are not "clean batches since")
        self.handoff_fallbacks += 1
        for _ in range(batches):
            self.run_batch()
        ops.check_coupled_status(dev)

    def stats(self, reset=True):
        """(episodes finished, mean of episode-mean reward, mean of episode-std, collisions) since last call."""
        self.flush()
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
        replayed: every evaluation starts from the same initial conditions and costs a graph replay instead of ~6 T eager launches
        and a new env.  (The reference's in-training test re-seeds with `seed - 1` where `seed` has been incremented by every reset
        before it, cacc_env.py:170-176: its test conditions move from episode to episode; here ONE fixed set of n_envs episodes,
        seed - 1 of the configured seed, serves as the yardstick of a run -- pass `seed` for another.)"""
        key = (int(n_envs), self.env.seed - 1 if seed is None else int(seed))
        cache = self.__dict__.setdefault('_eval_cache', {})
        if key not in cache:
            cache[key] = self._build_eval(*key)
        ev = cache[key]
        E0 = self.model.E
        ev['prepare']()
        if ev['graph'] is not None:
            ev['graph'].replay()
        else:
            ev['episode']()
        ev['statistics']()
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

        def prepare():
            env.episode.zero_()                       # the same test episode every time
            hs[0].zero_()
            cs[0].zero_()
            fps[0].copy_(model.fp_uniform.expand_as(fps[0]))

        def episode():
            # (captured: launches of this library only -- the state resets in front and the statistics behind run eagerly around
            # the replay.  With the aten fills / copies / reductions inside, a graph captured BEFORE the trainer's first batch
            # replayed its reductions on stale data once the trainer's own graphs existed: the episode's reward sums came out
            # right and the action histogram did not (seen in round 5's learning runs; tests/test_gpu_trainer.py pins it).)
            env.reset()
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

        def statistics():
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
                prepare()
                episode()                             # warm-up (allocator, library handles)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = self._new_graph()
            with torch.cuda.graph(graph):
                episode()
        return dict(env=env, prepare=prepare, episode=episode, statistics=statistics, graph=graph, hist=hist, total=total, steps=steps,
                    acts=acts, done=D)

    def run(self, log_every=10, eval_every=None):
        """Train until the counter says stop (`total_step` lock-steps per replica); one row per `log_every` batches.
        Row: `avg_reward` / `std_reward` = the deterministic TEST episodes (argmax policy, raw reward) for CACC, like
        the reference's train_reward.csv (utils.py:246-251), evaluated every `eval_every` rows -- default: every env.T / n_step
        rows (the reference tests once per training episode of ONE replica, utils.py:246-251; here a row already spans log_every
        batches of E replicas), at the FIRST row and at the last one; never for ATSC, whose logged
        reward is the training episode's (utils.py:243-245) -- and carried forward on the rows in between (`evaluated` marks the
        rows that ran one; the first row runs one, so no row is NaN); `train_avg_reward` etc. = statistics of the training episodes
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
                ran = final or rows_done == 0 or rows_done % eval_every == eval_every - 1
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
