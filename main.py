"""Entry point with the reference's CLI (main.py:21-40 of cts198859/deeprl_network):

    python main.py --base-dir D train --config-dir config/config_ia2c_fp_catchup.ini
    python main.py --base-dir D evaluate --evaluation-seeds 2000,2010

Optional batched keys: `num_envs` in [ENV_CONFIG] (or --num-envs) selects the MI355X batched
trainer (E lock-stepped replicas, Philox RNG); num_envs = 1 (default when absent) runs the
reference's single-replica loop with the global NumPy RNG.  Multi-GPU: launch with
`python -m torch.distributed.run --nproc-per-node N main.py ...` (one process per GPU, RCCL).
"""
from deeprl_network_amd.main import main

if __name__ == '__main__':
    main()
