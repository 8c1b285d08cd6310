"""profiles/rNN_untraced_breakdown.md from a bench.py JSON line: where the batch's time goes WITHOUT the tracer -- hipGraph timings
between HIP events on the launch stream, their differences, and (round 6) durations from device time stamps inside the captured graphs
(bench.py: measure_lstm_step_in_rollout, measure_lstm_step_stamped, update_breakdown), so that the numbers the line prints can be
recomputed from a tracked file.
    python tools/untraced_breakdown.py gpurun_out/r06_bench_default.json r06 > profiles/r06_untraced_breakdown.md"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
TAG = sys.argv[2] if len(sys.argv) > 2 else 'r06'


def rows(tag, o):
    r, u = o.get('roofline', {}), o.get('update', {})
    rb = o.get('roofline_bptt', {})
    ms = o['ms_per_step'] * 1e3
    out = ['### %s\n' % tag, '| item | us | how |', '|---|---:|---|',
           '| batch (timed region, host clock / steps) | %.1f | `ms_per_step` |' % ms]
    if r.get('rollout_graph_us'):
        n = r.get('launches_per_batch', 61)
        out += ['| rollout graph | %.1f | replay between HIP events |' % r['rollout_graph_us'],
                '| rollout graph without the lock-step launches | %.1f | same graph captured without them |' % r['rollout_graph_us_without_lstm_steps'],
                '| lock-step launch (`%s`) | %.2f | (full - without) / %d launches -> frac %.3f of the fp32 matrix peak |'
                % (r.get('kernel', '?')[:48], r['us_per_launch'], n, r.get('frac', float('nan')))]
        if r.get('us_per_launch_stamped_in_rollout'):
            out += ['| the same launch between two device time stamps inside the rollout graph | %.2f | median of 3 x %d; includes the two kernel boundaries next to the stamps -> frac %.3f |'
                    % (r['us_per_launch_stamped_in_rollout'], n, r.get('frac_stamped') or float('nan'))]
    elif 'us_per_launch' in r:
        out += ['| lock-step launch (`%s`) | %.2f | in-rollout, by graph difference -> frac %.3f%s |'
                % (r.get('kernel', '?')[:48], r['us_per_launch'], r.get('frac', float('nan')),
                   '' if not r.get('us_per_launch_stamped_in_rollout') else '; between two device time stamps: %.2f us' % r['us_per_launch_stamped_in_rollout'])]
    if 'update_graph_us' in u:
        out += ['| update graph | %.1f | replay between HIP events |' % u['update_graph_us'],
                '| update graph without the BPTT launch | %.1f | same update captured without `%s` |' % (u['update_graph_us_without_bptt'], u['bptt_entry']),
                '| BPTT launch inside the update | %.1f | DURATION between two device time stamps inside the captured update -> %.3f of the HBM peak '
                '(%.1f us back to back in isolation; marginal cost = update graph with - without it: %s us) |'
                % (u['bptt_us_in_update'], rb.get('frac', float('nan')), rb.get('us_per_launch_back_to_back', float('nan')),
                   '%.1f' % u['marginal_us_in_update'] if u.get('marginal_us_in_update') else 'n/a')]
        if r.get('rollout_graph_us'):
            out += ['| batch - rollout graph - update graph | %.1f | graph-launch gaps + what the two graphs lose by alternating (clock / power: each is timed above replayed on its own) |'
                    % (ms - r['rollout_graph_us'] - u['update_graph_us'])]
    return out + ['']


print('# Untraced time breakdown of a batch (round %s)\n' % TAG[1:].lstrip('0'))
print('Source: one `python bench.py --steps 20 --warmup 5` line (same process, same box as `%s_bench_default.json`); no profiler attached -- '
      'rocprofv3 serialises the graph\'s launches and reads ~5-10 %% longer per kernel (`%s_bench_kernel_stats.md` is the traced view).\n' % (TAG, TAG))
for line in rows('%s -- %.1f M env-steps/s' % (d['config']['workload'], d['value'] / 1e6), d):
    print(line)
for o in d.get('other_configs', []):
    if 'error' in o:
        continue
    for line in rows('%s -- %.1f M env-steps/s' % (o['workload'], o['value'] / 1e6), o):
        print(line)
