"""Data-parallel path ON THE GPU (the GPU twin of tests/test_dist_cpu.py).

A gpurun box has ONE MI355X, and RCCL (like NCCL) refuses two ranks on the same device ("Duplicate GPU
detected"), so:
  * with >= 2 visible devices the two-rank tests run on RCCL, one rank per device;
  * on a one-device box the two ranks share cuda:0 and exchange the flat gradient through gloo (device tensors,
    same product code: one all_reduce + 1/world inside the clip/RMSProp kernel) -- RCCL itself is exercised by the
    world-size-1 test (communicator creation, the all-reduce enqueued on the HIP stream between backward and the
    optimiser kernel) and by `bench.py` under NMARL_BENCH_FORCE_DIST=1.
The reference has no collective (models.py:34-42, 211-215 are single-process): the contract is that N ranks x E
replicas == one process x N*E replicas."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _backend_and_device(rank):
    if torch.cuda.device_count() >= 2:
        return 'nccl', rank
    return 'gloo', 0


def _run(agent, E, env_id_base, group, n_batches, device):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    from helpers import cacc_config
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    cp = cacc_config(agent=agent, n_step=10, reward_norm=800.0 if agent.startswith('ia2c') else 5000.0)
    cp['ENV_CONFIG']['episode_length_sec'] = '2'                   # T = 20 = 2 batches: episode ends + auto-reset inside
    env = CACCBatchEnv(cp['ENV_CONFIG'], num_envs=E, device=device, env_id_base=env_id_base)
    np.random.seed(12)
    cls = {'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC}[agent]
    model = cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, 10 ** 6,
                cp['MODEL_CONFIG'], seed=12, num_envs=E, device=device, dist_group=group)
    world = 1 if group is None else dist.get_world_size(group)
    tr = BatchedTrainer(env, model, Counter(10 ** 9, 10 ** 9, 10 ** 9), use_graph=True,
                        rank=0 if group is None else dist.get_rank(group), world_size=world)
    for _ in range(n_batches):
        tr.run_batch()
    torch.cuda.synchronize()
    return model.policy.params.flat.detach().cpu().clone(), tr.global_counter.cur_step


def _worker(rank, world, port, agent, E, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    backend, dev = _backend_and_device(rank)
    torch.cuda.set_device(dev)
    device = torch.device('cuda', dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    w, steps = _run(agent, E, rank * E, dist.group.WORLD, 3, device)
    torch.save((w, steps, backend), os.path.join(out, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('agent', ['ia2c_fp', 'ma2c_nc'])
def test_two_ranks_equal_one_process_on_gpu(agent, tmp_path):
    E = 64
    port = 29700 + (os.getpid() % 1500) + (0 if agent == 'ia2c_fp' else 11)
    mp.spawn(_worker, args=(2, port, agent, E, str(tmp_path)), nprocs=2, join=True)
    w0, s0, be = torch.load(tmp_path / 'rank0.pt')
    w1, s1, _ = torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(w0, w1), 'ranks diverged (%s)' % be
    assert s0 == s1 == 3 * 10
    single, _ = _run(agent, 2 * E, 0, None, 3, torch.device('cuda', 0))
    torch.testing.assert_close(w0, single, rtol=2e-4, atol=2e-6)


def _rccl_one_rank(rank, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)     # 'nccl' IS RCCL on ROCm
    w, _ = _run('ia2c_fp', 64, 0, dist.group.WORLD, 3, device)
    t = torch.arange(8, dtype=torch.float32, device=device)
    dist.all_reduce(t)
    torch.save((w, t.cpu()), os.path.join(out, 'rccl.pt'))
    dist.destroy_process_group()


def test_rccl_communicator_world1(tmp_path):
    """RCCL itself on this box: a one-rank communicator carries the product's gradient all-reduce; the result must be
    bit-identical to the run without a process group (sum over one rank, grad_scale 1)."""
    port = 29400 + (os.getpid() % 1500)
    mp.spawn(_rccl_one_rank, args=(port, str(tmp_path)), nprocs=1, join=True)
    w, t = torch.load(tmp_path / 'rccl.pt')
    assert torch.equal(t, torch.arange(8, dtype=torch.float32))
    plain, _ = _run('ia2c_fp', 64, 0, None, 3, torch.device('cuda', 0))
    assert torch.equal(w, plain)


def _bench(extra_env, *argv):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env, capture_output=True,
                       text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    return p, lines


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment (how the driver may call it) spawns its own two
    ranks and rank 0 prints exactly one JSON line with n_gpus = 2."""
    hooks = {} if torch.cuda.device_count() >= 2 else {'NMARL_BENCH_ONE_DEVICE': '1', 'NMARL_DIST_BACKEND': 'gloo'}
    p, lines = _bench(hooks, '--gpus', '2', '--steps', '2', '--warmup', '1', '--envs', '256', '--no-cpu-baseline')
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_replicas'] == 512 and d['scaling'] == 'weak'
    assert d['value'] > 0 and 'roofline' in d


def test_bench_self_launches_eight_ranks():
    """The driver's 8-GPU scaling run without the hardware: `python bench.py --gpus 8` on whatever devices there are
    (8 ranks on one device through gloo when fewer than 8 are visible): self_launch, the one-time GEMM tuning by rank 0
    with seven ranks waiting at the barrier, rank-offset replica ids, eight-way gradient all-reduce.  Rank 0 prints ONE
    line that explains itself: n_gpus, per-rank ms per step, the all-reduce time."""
    hooks = {} if torch.cuda.device_count() >= 8 else {'NMARL_BENCH_ONE_DEVICE': '1', 'NMARL_DIST_BACKEND': 'gloo'}
    p, lines = _bench(hooks, '--gpus', '8', '--steps', '2', '--warmup', '1', '--envs', '128', '--no-cpu-baseline')
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['config']['global_replicas'] == 8 * 128 and d['config']['parallelism'] == 'dp8'
    assert len(d['per_rank_ms_per_step']) == 8 and max(d['per_rank_ms_per_step']) == pytest.approx(d['ms_per_step'], rel=1e-6)
    ar = d['grad_allreduce']
    assert ar['per_update'] == 1 and ar['us'] > 0 and ar['bytes'] > 1e6


def test_bench_one_rank_on_rccl():
    """bench.py with the process group forced on for one rank: communicator + all-reduce on RCCL in the timed loop."""
    p, lines = _bench({'NMARL_BENCH_FORCE_DIST': '1'}, '--steps', '2', '--warmup', '1', '--envs', '256',
                      '--no-cpu-baseline')
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 1
