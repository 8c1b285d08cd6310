"""Shared test helpers (CPU-safe)."""
import configparser
import io
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CACC_INI = """
[MODEL_CONFIG]
rmsp_alpha = 0.99
rmsp_epsilon = 1e-5
max_grad_norm = 40
gamma = 0.99
lr_init = 5e-4
lr_decay = constant
entropy_coef = 0.05
value_coef = 0.5
num_lstm = 64
num_fc = 64
batch_size = {n_step}
reward_norm = {reward_norm}
reward_clip = -1

[TRAIN_CONFIG]
total_step = {total_step}
test_interval = 2e6
log_interval = 1e4

[ENV_CONFIG]
control_interval_sec = 0.1
episode_length_sec = 60
agent = {agent}
batch_size = {n_step}
coop_gamma = {coop_gamma}
headway_min = 1
headway_st = 5
headway_go = 35
speed_max = 30
accel_max = 2.5
accel_min = -2.5
reward_v = 1
reward_u = 0.1
collision_penalty = 1000
headway_target = 20
speed_target = 15
norm_headway = 10
norm_speed = 7.5
n_vehicle = 8
scenario = cacc_{scenario}
seed = {seed}
test_seeds = 10000,20000
"""


def cacc_config(agent='ma2c_nc', scenario='catchup', seed=12, coop_gamma=-1, n_step=60,
                reward_norm=5000.0, total_step=1200):
    cp = configparser.ConfigParser()
    cp.read_file(io.StringIO(CACC_INI.format(agent=agent, scenario=scenario, seed=seed,
                                             coop_gamma=coop_gamma, n_step=n_step,
                                             reward_norm=reward_norm, total_step=total_step)))
    return cp


def load_npz(path):
    with np.load(path, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}
