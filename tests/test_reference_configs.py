"""Drop-in of the ini schema: every CACC / large-grid config shipped by the REFERENCE parses unchanged into the
kernel parameter structs, and the matching model class constructs with the reference's parameter count.
Needs the reference checkout (authoring container); skipped on the GPU box."""
import configparser
import glob
import os

import numpy as np
import pytest

REF_CFG = '/root/reference/config'
FILES = sorted(glob.glob(os.path.join(REF_CFG, '*.ini')))
# parameter counts of the reference's graphs (SURVEY.md 8a + tests/golden/nn_*.npz)
PARAMS = {('ia2c', 8): 274400, ('ia2c_fp', 8): 409568, ('ma2c_nc', 8): 598496, ('ma2c_ic3', 8): 307680,
          ('ma2c_cu', 8): 269920, ('ma2c_dial', 8): 365536, ('ma2c_ic3', 25): 1021990, ('ma2c_nc', 25): 2093670,
          ('ma2c_cu', 25): 856550, ('ma2c_dial', 25): 1351270}

pytestmark = pytest.mark.skipif(not FILES, reason='reference checkout not present')


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f) for f in FILES])
def test_reference_ini_parses_and_model_builds(path):
    from cpu_emulation import cpu_ops
    from deeprl_network_amd.agents import models
    from deeprl_network_amd.envs.cacc_env import _params_from_config, line_graph
    from deeprl_network_amd.envs.large_grid_env import grid_masks, grid_params_from_config
    cp = configparser.ConfigParser()
    cp.read(path)
    env_cfg = cp['ENV_CONFIG']
    scenario, agent = env_cfg.get('scenario'), env_cfg.get('agent')
    if scenario == 'atsc_real_net':
        pytest.skip('Monaco net (SUMO) is out of scope: SURVEY.md section 2 row 11')
    if agent == 'greedy':
        pytest.skip('greedy controller config has no learner')
    mc = cp['MODEL_CONFIG']
    if scenario.startswith('cacc'):
        p, name = _params_from_config(env_cfg)
        assert p.T == 600 and p.batch_size == mc.getint('batch_size') and name in ('catchup', 'slowdown')
        nb, dist = line_graph(env_cfg.getint('n_vehicle'))
        n_feat, A = 5, 4
    else:
        p = grid_params_from_config(env_cfg)
        assert p.T == 720 and p.T % mc.getint('batch_size') == 0
        nb, dist = grid_masks()
        n_feat, A = 12, 5
    N = nb.shape[0]
    is_ma = agent.startswith('ma2c')
    n_s_ls = [n_feat if is_ma else n_feat * (1 + int(nb[i].sum())) for i in range(N)]
    cls = {'ia2c': models.IA2C, 'ia2c_fp': models.IA2C_FP, 'ma2c_nc': models.MA2C_NC, 'ma2c_ic3': models.MA2C_IC3,
           'ma2c_cu': models.IA2C_CU, 'ma2c_dial': models.MA2C_DIAL}[agent]
    total_step = int(cp.getfloat('TRAIN_CONFIG', 'total_step'))
    np.random.seed(env_cfg.getint('seed'))
    with cpu_ops():
        model = cls(n_s_ls, [A] * N, nb, dist, env_cfg.getfloat('coop_gamma'), total_step, mc,
                    seed=env_cfg.getint('seed'), num_envs=1, device='cpu')
    n_params = sum(a.size for _, a in model.policy.params.ref_variables())
    if (agent, N) in PARAMS:
        assert n_params == PARAMS[(agent, N)]
    assert model.n_step == mc.getint('batch_size') and model.coop_gamma == env_cfg.getfloat('coop_gamma')
