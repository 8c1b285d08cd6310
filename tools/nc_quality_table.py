"""profiles/r03_nc_quality.md: greedy-test quality of MA2C-NeurComm on CACC slow-down over the reference's full schedule
(1e6 lock-steps per replica = 16 667 updates) -- the batched product at E = 8 ... 4096 (tools/nc_quality.sh, logs under
gpurun_out/), the E = 1 CPU port of the reference loop (tests/learning/port_full_schedule.py, pinned step for step to the
real reference over three episodes by tests/test_e2e_multi_cpu.py) and the reference's own published numbers.
    python tools/nc_quality_table.py > profiles/r03_nc_quality.md"""
import glob
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_of(path):
    out = []
    for ln in open(path):
        if ln.startswith('{'):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def product_runs():
    runs = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'nc_quality_E*_s*.log')),
                    key=lambda p: [int(x) for x in re.findall(r'\d+', os.path.basename(p))]):
        E, s = [int(x) for x in re.findall(r'\d+', os.path.basename(f))][:2]
        r = [x for x in rows_of(f) if x.get('batch')]
        if r:
            done_ = r[-1]['batch'] >= 16000
            runs.append((E, s, r, 'this round' if done_ else 'this round, first attempt: 7 concurrent processes on the GPU, cut by the time limit'))
    old = os.path.join(ROOT, 'profiles', 'r02_learn_ma2c_nc_slowdown.json')
    if os.path.exists(old):
        d = json.load(open(old))
        runs.append((d['E'], 12, [x for x in d['rows'] if x.get('batch')], 'round 2 (profiles/r02_learn_ma2c_nc_slowdown.json)'))
    return runs


READING = '''
## Reading

* **The reference's own loop does not reach its notebook numbers under its shipped configuration.**  `config_ma2c_nc_slowdown.ini`
  (every MODEL / ENV key identical to the config used here; checked key by key) through the E = 1 port -- which replays the real
  reference Trainer + CACCEnv + MA2C_NC action for action over three training and test episodes, so it IS the reference algorithm
  up to float32-vs-float64 rounding -- ends the full 1e6-step schedule with a greedy test reward of about -2500 and 42-50 collisions
  in its last 50 test episodes, both seeds.  Training episodes (stochastic policy) improve steadily (collisions 73 % -> 33 %),
  the ARGMAX policy the reference logs for CACC does not.  The notebook's -894 / 13 collisions were therefore produced by something
  other than this code + ini (the cells were re-run per scenario and the outputs are stale, BASELINE.md).
* **The batched product behaves the same way, only better with more replicas.**  Same algorithm, same hyper-parameters, mean gradient over
  E replicas: training collisions fall from 20 % of the episodes (E = 8) over 1.2 % (E = 64) and 0.2 % (E = 512) to 0.04 % (E = 4096) in the last quarter, the greedy test
  policy still collides in most test episodes at every E (E = 64, seed 12 dips to 3 / 64 at one evaluation and is back at 64 / 64 at
  others: the argmax of a high-entropy policy -- entropy coefficient 0.05 -- flips between action patterns from one evaluation to the
  next).  There is no trend with E that would point at the batch seam: E = 8 is as bad as E = 1, E >= 64 is better than E = 1.
* **So the gap VERDICT r2 saw is not a defect of the batched path** (episode seam, state / fingerprint reset, evaluate()): the seam is
  additionally pinned against the real reference over three episodes (tests/golden/e2e_multi_ma2c_nc_slowdown.npz) on CPU and GPU.
  What it is: the slow-down task starts every vehicle 50-150 % too fast; the stochastic policy brakes by mixing the four (alpha, beta)
  gains, its argmax commits to one of them -- avoiding collisions greedily needs a much lower-entropy policy than 1e6 steps at
  e_coef 0.05 produce.  (IA2C-FP catch-up, the bench workload, trains to 0 greedy collisions: profiles/r02_learning_curves.md.)
'''


def main():
    print('# NeurComm slow-down: greedy-test quality over the full schedule (round 3)\n')
    print('Question (VERDICT r2, weak #1): after the full schedule the batched product\'s GREEDY NeurComm policy collides in 41-62 of 64 '
          'test episodes while `result_plot.ipynb:746-781` lists 13 / 50 for the reference -- a defect of the batched path '
          '(episode seam, state reset, fingerprint reset, evaluate()) or the algorithm\'s own behaviour on this task?\n')
    print('Method: the SAME product code at E = 8, 64, 512, 4096 replicas (`tools/nc_quality.sh`: `tools/learn_curve.py ma2c_nc slowdown E 16667 500 seed`, '
          'ini defaults: lr 5e-4 constant, RMSProp, clip 40, n_step 60; greedy test = 64 argmax episodes with the raw reward every 500 updates), '
          'and the reference\'s own loop at E = 1 on the CPU port (`tests/learning/port_full_schedule.py`: the test episode after every training '
          'episode like `utils.py:246-251`; the port replays the REAL reference -- env, model and Trainer on the TF shim -- action for action '
          'over three training + test episodes, `tests/test_e2e_multi_cpu.py`).\n')
    print('## Batched product (MI355X), last quarter of the schedule (updates 12 500 - 16 667)\n')
    print('| E | seed | updates done | train avg r | train collisions / episodes | greedy test avg r | greedy test collisions / 64 (min-max over the rows) | source |')
    print('|---:|---:|---:|---:|---|---:|---|---|')
    for E, s, r, src in product_runs():
        last = r[-1]['batch']
        tail = [x for x in r if x['batch'] > 0.75 * 16667] or r[-3:]
        tc = [x['test_collisions'] for x in tail]
        print('| %d | %d | %d | %.1f | %d / %d | %.1f | %d-%d (mean %.1f) | %s |' % (
            E, s, last, np.mean([x['train_avg_reward'] for x in tail]), sum(x['train_collisions'] for x in tail),
            sum(x['train_episodes'] for x in tail), np.mean([x['test_avg_reward'] for x in tail]), min(tc), max(tc), np.mean(tc), src))
    print('\n## E = 1, the reference loop itself (CPU port), by quarter of the schedule\n')
    print('| seed | lock-steps | training episodes | train avg r | train collisions | greedy test avg r (the value the reference logs) | greedy test collisions |')
    print('|---:|---|---:|---:|---|---:|---|')
    for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'port', 'nc_slowdown_seed*.jsonl'))):
        s = int(re.findall(r'\d+', os.path.basename(f))[0])
        r = rows_of(f)
        for lo, hi in ((0, 250000), (250000, 500000), (500000, 750000), (750000, 10 ** 9)):
            w = [x for x in r if lo < x['step'] <= hi]
            if not w:
                continue
            print('| %d | %d-%d | %d | %.1f | %d / %d | %.1f | %d / %d |' % (
                s, lo, min(hi, w[-1]['step']), len(w), np.mean([x['train_avg_reward'] for x in w]), sum(x['train_collision'] for x in w), len(w),
                np.mean([x['avg_reward'] for x in w]), sum(x['test_collision'] for x in w), len(w)))
        last = r[-50:]
        print('| %d | last 50 episodes (to %d) | 50 | %.1f | %d / 50 | **%.1f** | **%d / 50** |' % (
            s, r[-1]['step'], np.mean([x['train_avg_reward'] for x in last]), sum(x['train_collision'] for x in last),
            np.mean([x['avg_reward'] for x in last]), sum(x['test_collision'] for x in last)))
    print('\n## Reference, published (stale notebook outputs, BASELINE.md)\n')
    print('* `result_plot.ipynb:340-345` (notebook set to CACC slow-down): NeurComm train log, avg R of the last 50 episodes (= the greedy test '
          'episodes `utils.py:246-251` logs) **-894.51**.')
    print('* `result_plot.ipynb:746-781` (execution over 50 seeds, scenario attribution ambiguous): ma2c_nc **-934.73, 13 / 50 collisions**.')
    print(READING)


if __name__ == '__main__':
    main()
