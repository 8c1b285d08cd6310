"""Turn the raw outputs of `tools/prof_all.sh <tag>` (gpurun_out/<tag>_*) into the committed summaries under profiles/:
<tag>_bench_default.json, <tag>_other_configs.md, <tag>_bench_kernel_stats.md, <tag>_pmc_traffic.{json,md}, <tag>_step_timeline.txt.
    python tools/profiles_summary.py r03"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out')
DST = os.path.join(ROOT, 'profiles')

# (config file stem, row label, agents x replicas, round-2 M/s, first round-3 measurement M/s: profiles history)
ROWS = [('default', 'IA2C-FP catch-up (configs[1], bench default)', '8 x 4096', 210.4, 218.5),
        ('ma2c_nc_slowdown', 'NeurComm slow-down (configs[2])', '8 x 4096', 111.1, 133.5),
        ('ma2c_cnet_grid', 'CommNet, synthetic 5x5 grid (configs[3])', '25 x 1024', 109.1, 135.6),
        ('ma2c_nc_catchup', 'NeurComm catch-up (configs[4] per GPU)', '8 x 4096', 111.1, 133.2),
        ('ma2c_cnet_catchup', 'CommNet catch-up', '8 x 4096', 149.3, 183.4),
        ('ma2c_dial_catchup', 'DIAL catch-up', '8 x 4096', 86.1, 85.3),
        ('ia2c_cu_catchup', 'ConseNet catch-up', '8 x 4096', 259.4, 264.6)]
SECTIONS = [('ia2c_fp_catchup', 'IA2C-FP catch-up, 8 x 4096 (BASELINE configs[1], the bench default)'),
            ('ma2c_nc_slowdown', 'NeurComm slow-down, 8 x 4096 (BASELINE configs[2])'),
            ('ma2c_cnet_grid', 'CommNet on the synthetic 5x5 grid, 25 x 1024 (BASELINE configs[3])')]


def last_json(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def main(tag):
    d0 = last_json(os.path.join(SRC, '%s_bench_default.json' % tag))
    with open(os.path.join(DST, '%s_bench_default.json' % tag), 'w') as f:
        json.dump(d0, f, indent=1)
        f.write('\n')
    out = ['# All benchmarked configs (round %s)\n' % tag.lstrip('r0'),
           '`python bench.py --no-cpu-baseline --steps 10 --config config/config_<cfg>.ini` on one MI355X (`tools/prof_all.sh %s`; the '
           'default row: `python bench.py --steps 20 --warmup 5`); agent-steps/s = agents x replicas x lock-steps / s over rollout + '
           'update.  `us / launch` = the LSTM lock-step launch inside the rollout (difference of two un-profiled rollout-graph '
           'timings, bench.py `roofline.how`).  "first" = the first measurement pass of this round (before the one-launch coupled '
           'step, the permuted column tiles and the in-kernel grid encoder).  The default line was sampled on six boxes of the pool '
           'this round (same code path): 8.93, 8.99, 9.00, 9.06, 9.10, 9.19, 9.27, 9.37, 9.38 ms = 209.5 ... 220.1 M env-steps/s; '
           'under the tracer the batch spans 9.23 - 9.28 ms on every box -- the table holds the LAST sample.\n' % tag,
           '| config | agents x replicas | round 2 M/s | round 3 first M/s | now M/s | ms / batch | LSTM lock-step kernel | us / launch | frac of fp32 MFMA peak |',
           '|---|---|---:|---:|---:|---:|---|---:|---:|']
    for stem, label, shape, r2, first in ROWS:
        d = d0 if stem == 'default' else last_json(os.path.join(SRC, '%s_bench_%s.json' % (tag, stem)))
        r = d.get('roofline', {})
        us = r.get('us_per_launch_in_rollout') or r.get('us_per_launch')
        out.append('| %s | %s | %.1f | %.1f | **%.1f** | %.2f | %s | %.1f | %.2f |' % (
            label, shape, r2, first, d['value'] / 1e6, d['ms_per_step'], r.get('kernel', '?').split(' (')[0], us, r.get('frac', float('nan'))))
    out.append('\nSide measurements of the default run (bench.py JSON, profiles/%s_bench_default.json):\n' % tag)
    for k in sorted(d0):
        r = d0[k]
        if k.startswith('roofline') and isinstance(r, dict) and 'frac' in r:
            tr = r.get('traffic')
            out.append('* `%s`: %s -- %.1f us/launch, %.1f %s = %.3f of the %s roofline; PMC traffic %s' % (
                k, r.get('kernel', '?').split(' (')[0], r.get('us_per_launch_in_rollout') or r['us_per_launch'], r['achieved'], r['unit'], r['frac'],
                r['bound'], 'n/a' if tr is None else '%.1f MB/launch (%s)' % (tr / 1e6, r.get('traffic_source'))))
    cb = d0.get('cpu_baseline')
    if cb:
        out.append('* `cpu_baseline` (%s): %.0f env-steps/s on %d core of the bench host (%s, %d cores); %d processes: %.0f; '
                   'the port of the GPU workload: %.0f' % (cb.get('config'), cb['value'], cb['cores'], cb.get('host_cpu'), cb.get('host_cores', 0),
                                                            cb.get('all_cores', {}).get('cores', 0), cb.get('all_cores', {}).get('value', 0),
                                                            cb.get('workload_matched', {}).get('value', 0)))
    with open(os.path.join(DST, '%s_other_configs.md' % tag), 'w') as f:
        f.write('\n'.join(out) + '\n')
    ks = ['# rocprofv3 --kernel-trace --stats summaries (round %s)' % tag.lstrip('r0'),
          'command (per config, after one untraced run that writes the TunableOp GEMM choices): `rocprofv3 --kernel-trace --stats -d '
          '/tmp/prof_<cfg> -o bench -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 --config config/config_<cfg>.ini`, then '
          '`python tools/rocpd_stats.py /tmp/prof_<cfg>/bench_results.db --steps 7 --update` for the per-batch view '
          '(`tools/prof_cfg.sh`, `tools/prof_all.sh %s`).  The `calls=` column counts every launch of the traced process: the 7 '
          'batches AND bench.py\'s roofline side measurements (e.g. the 2^21-replica env step, the 60-launch LSTM graphs), so `/7` is '
          'an upper bound for those kernels; the per-batch line at the end of each block is exact.  The kernel trace serialises the '
          'launches of a captured graph: per-kernel durations inside the rollout read ~10 %% longer than in the un-profiled run -- '
          '`bench.py` therefore quotes the LSTM lock-step by difference of two un-profiled graph timings (`roofline.how`).\n' % tag]
    for stem, title in SECTIONS:
        p = os.path.join(SRC, '%s_stats_%s.txt' % (tag, stem))
        if not os.path.exists(p):
            continue
        ks += ['## ' + title, '', '```', open(p).read().rstrip(), '```', '']
    with open(os.path.join(DST, '%s_bench_kernel_stats.md' % tag), 'w') as f:
        f.write('\n'.join(ks) + '\n')
    for name in ('pmc_traffic.json', 'pmc_traffic.md', 'step_timeline.txt'):
        p = os.path.join(SRC, '%s_%s' % (tag, name))
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, '%s_%s' % (tag, name)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r03')
