"""Environment factory (main.py:43-51 of the reference)."""


def make_batch_env(config, num_envs, device='cuda', seed=None, env_id_base=0):
    """ENV_CONFIG section -> batched device-resident environment."""
    scenario = config.get('scenario')
    if scenario.startswith('atsc') or scenario == 'large_grid':          # (config_greedy.ini names the scenario `large_grid`)
        if scenario.endswith('large_grid'):
            from .large_grid_env import LargeGridBatchEnv
            return LargeGridBatchEnv(config, num_envs=num_envs, device=device, seed=seed, env_id_base=env_id_base)
        from .real_net_env import RealNetBatchEnv
        return RealNetBatchEnv(config, num_envs=num_envs, device=device, seed=seed, env_id_base=env_id_base)
    from .cacc_env import CACCBatchEnv
    return CACCBatchEnv(config, num_envs=num_envs, device=device, seed=seed, env_id_base=env_id_base)


def init_env(config, port=0, device='cuda'):
    """Single-replica env with the reference duck-type (main.py:43-51)."""
    scenario = config.get('scenario')
    if scenario.startswith('atsc') or scenario == 'large_grid':
        if scenario.endswith('large_grid'):
            from .large_grid_env import LargeGridEnv
            return LargeGridEnv(config, port=port, device=device)
        from .real_net_env import RealNetEnv
        return RealNetEnv(config, port=port, device=device)
    from .cacc_env import CACCEnv
    return CACCEnv(config, device=device)
