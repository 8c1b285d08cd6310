"""Markdown tables of full-schedule learning runs (tools/learn_curve.py's JSON rows as printed to its log):
    python tools/learn_table.py <title>=<log> [<title>=<log> ...] > profiles/rNN_learning_curves.md"""
import json
import sys

print('# Full-schedule learning runs on the round-6 kernels (one launch per lock-step incl. the grid\'s env step, carried message term, '
      'fused heads + loss pass, captured update)')
print('command: `python tools/learn_curve.py <agent> <scenario | ini> <E> <updates> 1000` on one MI355X (n_step and every other setting: the ini '
      'defaults; updates = the reference\'s `total_step = 1e6` lock-steps / n_step).  `train_*` = training episodes finished since the previous row '
      '(stochastic policy, training reward); `test_*` = 64 deterministic argmax episodes, raw reward (utils.py:246-251).  Round 5\'s runs of the '
      'same schedules: `profiles/r05_learning_curves.md` -- the curves are not bit-comparable (the fused heads + loss pass adds the 64-long head '
      'dots and the weight-gradient partial sums in another order), the levels they reach are.\n')
for arg in sys.argv[1:]:
    title, path = arg.rsplit('=', 1)
    rows = [json.loads(l) for l in open(path) if l.startswith('{')]
    print('## %s\n' % title)
    print('| update | train episodes | train avg r | train collisions | test avg r | test std | test collisions / 64 | greedy action share | wall s |')
    print('|---:|---:|---:|---:|---:|---:|---:|---|---:|')
    for r in rows:
        f = lambda k, fmt='%.1f': '' if r.get(k) is None else (fmt % r[k])      # noqa: E731
        print('| %d | %s | %s | %s | %s | %s | %s | %s | %s |' % (
            r['batch'], f('train_episodes', '%d'), f('train_avg_reward'), f('train_collisions', '%d'), f('test_avg_reward'), f('test_std_reward'),
            f('test_collisions', '%d'), [round(x, 4) for x in r.get('test_action_share', [])], f('wall_s')))
    print()
