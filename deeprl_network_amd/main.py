"""train / evaluate drivers -- the reference's main.py (21-163) on the MI355X path."""
import argparse
import configparser
import logging
import os

import numpy as np
import torch

from .agents.models import IA2C, IA2C_CU, IA2C_FP, MA2C_DIAL, MA2C_IC3, MA2C_NC
from .envs import init_env, make_batch_env
from .envs.large_grid_env import LargeGridController
from .utils import (BatchedTrainer, Counter, Evaluator, SummaryWriter, Trainer, check_dir, copy_file, find_file,
                    init_dir, init_log)

AGENTS = {'ia2c': IA2C, 'ia2c_fp': IA2C_FP, 'ma2c_nc': MA2C_NC, 'ma2c_ic3': MA2C_IC3, 'ma2c_cu': IA2C_CU,
          'ma2c_dial': MA2C_DIAL}


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--base-dir', type=str, required=False, default='./runs/default', help='experiment base dir')
    subparsers = parser.add_subparsers(dest='option', help='train or evaluate')
    sp = subparsers.add_parser('train', help='train a single agent under base dir')
    sp.add_argument('--config-dir', type=str, required=False, default='./config/config_ia2c_fp_catchup.ini',
                    help='experiment config path')
    sp.add_argument('--num-envs', type=int, default=None, help='replicas per GPU (overrides ENV_CONFIG num_envs)')
    sp.add_argument('--no-graph', action='store_true', help='do not capture the rollout in a hipGraph')
    sp = subparsers.add_parser('evaluate', help='evaluate and compare agents under base dir')
    sp.add_argument('--evaluation-seeds', type=str, required=False,
                    default=','.join([str(i) for i in range(2000, 2500, 10)]),
                    help='random seeds for evaluation, split by ,')
    sp.add_argument('--demo', action='store_true', help='kept for CLI compatibility (no SUMO gui here)')
    args = parser.parse_args(argv)
    if not args.option:
        parser.print_help()
        raise SystemExit(1)
    return args


def init_agent(env, config, total_step, seed, **kw):
    if env.agent == 'greedy':                  # rule-based baseline of the ATSC scenarios (large_grid_env.py:30-45)
        return LargeGridController(getattr(env, 'node_names', None)) if env.name.endswith('large_grid') else None
    cls = AGENTS.get(env.agent)
    if cls is None:
        return None
    return cls(env.n_s_ls, env.n_a_ls, env.neighbor_mask, env.distance_mask, env.coop_gamma, total_step, config,
               seed=seed, n_feat_ls=getattr(env, 'n_feat_ls', None), obs_order=getattr(env, 'neighbor_order', None), **kw)


def _dist():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    group = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        group = dist.group.WORLD
    return world, rank, local, group


def train(args):
    world, rank, local, group = _dist()
    dirs = init_dir(args.base_dir)
    init_log(dirs['log'], rank)
    if rank == 0:
        copy_file(args.config_dir, dirs['data'])
    config = configparser.ConfigParser()
    config.read(args.config_dir)
    env_cfg = config['ENV_CONFIG']
    total_step = int(config.getfloat('TRAIN_CONFIG', 'total_step'))
    test_step = int(config.getfloat('TRAIN_CONFIG', 'test_interval'))
    log_step = int(config.getfloat('TRAIN_CONFIG', 'log_interval'))
    counter = Counter(total_step, test_step, log_step)
    seed = config.getint('ENV_CONFIG', 'seed')
    num_envs = args.num_envs if args.num_envs is not None else env_cfg.getint('num_envs', fallback=1)
    device = torch.device('cuda', local)
    writer = SummaryWriter(dirs['log']) if rank == 0 else None
    if env_cfg.get('agent') == 'greedy':
        # the rule-based ATSC baseline has nothing to train (no add_transition / backward / save): say so instead of dying on an
        # AttributeError after the first env step with a half-initialised run directory -- `main.py evaluate` is its mode
        logging.error('Training: agent "greedy" is a rule-based controller; run `main.py evaluate` on it')
        return
    if num_envs <= 1 and world == 1:
        env = init_env(env_cfg, device=device)                       # seeds np.random (cacc_env.py:22)
        logging.info('Training: a dim %r, agent dim: %d' % (env.n_a_ls, env.n_agent))
        model = init_agent(env, config['MODEL_CONFIG'], total_step, seed, device=device)
        trainer = Trainer(env, model, counter, writer, output_path=dirs['data'])
        trainer.run()
    else:
        env = make_batch_env(env_cfg, num_envs=num_envs, device=device, env_id_base=rank * num_envs)
        np.random.seed(seed)                                         # same initial weights on every rank
        model = init_agent(env, config['MODEL_CONFIG'], total_step, seed, num_envs=num_envs, device=device,
                           dist_group=group)
        trainer = BatchedTrainer(env, model, counter, writer, output_path=dirs['data'],
                                 use_graph=not args.no_graph, rank=rank, world_size=world)
        trainer.run()
    if rank == 0:
        final_step = counter.cur_step
        logging.info('Training: save final model at step %d ...' % final_step)
        model.save(dirs['model'], final_step)


def _open_run(run_dir):
    """A finished training run on disk -> the parsed ini `train` left under <run>/data/ (main.py:112-123 of the reference finds it
    the same way), or None with the reason logged."""
    if not check_dir(run_dir):
        logging.error('Evaluation: %s does not exist!' % os.path.basename(run_dir.rstrip('/')))
        return None
    ini = find_file(os.path.join(run_dir, 'data') + '/')
    if not ini:
        return None
    cfg = configparser.ConfigParser()
    cfg.read(ini)
    return cfg


def evaluate(args):
    """`main.py evaluate` (reference main.py:112-155): the run under --base-dir is rebuilt from its own ini, its newest checkpoint
    loaded, and the Evaluator replays the test seeds into <run>/eva_data (port 1, no GUI: the synthetic envs have neither)."""
    run_dir = args.base_dir
    out = init_dir(run_dir, pathes=['eva_data', 'eva_log'])
    init_log(out['eva_log'])
    logging.info('Evaluation: random seeds: %s' % args.evaluation_seeds)
    seeds = [int(tok) for tok in args.evaluation_seeds.split(',')] if args.evaluation_seeds else []
    cfg = _open_run(run_dir)
    if cfg is None:
        return
    env = init_env(cfg['ENV_CONFIG'], port=1)
    env.init_test_seeds(seeds)
    model = init_agent(env, cfg['MODEL_CONFIG'], 0, 0)
    if model is None or not model.load(os.path.join(run_dir, 'model') + '/'):
        return
    Evaluator(env, model, out['eva_data'], gui=False).run()


def main(argv=None):
    args = parse_args(argv)
    if args.option == 'train':
        train(args)
    else:
        evaluate(args)
