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
            runs.append((E, s, r, 'this round'))
    old = os.path.join(ROOT, 'profiles', 'r02_learn_ma2c_nc_slowdown.json')
    if os.path.exists(old):
        d = json.load(open(old))
        runs.append((d['E'], 12, [x for x in d['rows'] if x.get('batch')], 'round 2 (profiles/r02_learn_ma2c_nc_slowdown.json)'))
    return runs


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


if __name__ == '__main__':
    main()
