"""Throughput of `main.py train` (BatchedTrainer.run: rollout graph + update + logging rows + in-training test episodes) next to
bench.py's bare run_batch loop, on the same config: steady-state env-steps/s between the row a third into the run and the last
row (the first rows contain the graph capture and the GEMM tuning).  python tools/train_speed.py [config.ini] [batches=300]"""
import configparser
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'config', 'config_ia2c_fp_catchup.ini')
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cp = configparser.ConfigParser()
cp.read(cfg)
n_step = cp.getint('MODEL_CONFIG', 'batch_size')
cp['TRAIN_CONFIG']['total_step'] = str(batches * n_step)
with tempfile.TemporaryDirectory() as d:
    ini = os.path.join(d, os.path.basename(cfg))
    with open(ini, 'w') as f:
        cp.write(f)
    env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_FILENAME='/tmp/nmarl_tunableop_%d.csv')
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'main.py'), '--base-dir', os.path.join(d, 'run'), 'train',
                           '--config-dir', ini], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import pandas as pd
    df = pd.read_csv(os.path.join(d, 'run', 'data', 'train_reward.csv'))
a, b = df.iloc[len(df) // 3], df.iloc[-1]
rate = (b['env_steps'] - a['env_steps']) / (b['wall_s'] - a['wall_s'])
print('main.py train, %s: %d rows, %d evaluated; steady state %.1f M env-steps/s (%.2f ms per n_step batch) between rows %d and %d'
      % (os.path.basename(cfg), len(df), int(df.get('evaluated', pd.Series([0])).sum()), rate / 1e6,
         (b['wall_s'] - a['wall_s']) / ((b['step'] - a['step']) / n_step) * 1e3, len(df) // 3, len(df) - 1))
