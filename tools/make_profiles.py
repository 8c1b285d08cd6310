"""Turn the scratch outputs of tools/prof_all.sh (gpurun_out/<tag>_*) into the tracked summaries under profiles/:
  <tag>_bench_default.json, <tag>_bench_kernel_stats.md, <tag>_other_configs.md, <tag>_pmc_traffic.{json,md},
  <tag>_pmc_lstm_stalls.md, <tag>_time_fused.txt, <tag>_env_microbench.txt.
  python tools/make_profiles.py [tag=r02] [src=gpurun_out]"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src = os.path.join(ROOT, sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out')
dst = os.path.join(ROOT, 'profiles')

CONFIGS = [('ia2c_fp_catchup', 'IA2C-FP catch-up, 8 x 4096 (BASELINE configs[1], the bench default)'),
           ('ma2c_nc_slowdown', 'NeurComm slow-down, 8 x 4096 (configs[2])'),
           ('ma2c_cnet_grid', 'CommNet on the synthetic 5x5 grid, 25 x 1024 (configs[3])'),
           ('ma2c_dial_catchup', 'DIAL catch-up, 8 x 4096 (outside BASELINE\'s configs; SURVEY 8f)')]
ROUND1 = {'default': 132.4, 'ma2c_nc_slowdown': 66.0, 'ma2c_cnet_grid': 67.2, 'ma2c_cnet_catchup': 88.2, 'ma2c_dial_catchup': 60.1,
          'ia2c_cu_catchup': 157.6}
LABEL = {'default': 'IA2C-FP catch-up (configs[1], bench default)', 'ma2c_nc_slowdown': 'NeurComm slow-down (configs[2])',
         'ma2c_cnet_grid': 'CommNet, synthetic 5x5 grid (configs[3])', 'ma2c_nc_catchup': 'NeurComm catch-up (configs[4] per GPU)',
         'ma2c_cnet_catchup': 'CommNet catch-up', 'ma2c_dial_catchup': 'DIAL catch-up', 'ia2c_cu_catchup': 'ConseNet catch-up'}


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


# ---- bench_default.json
shutil.copy(os.path.join(src, '%s_bench_default.json' % tag), os.path.join(dst, '%s_bench_default.json' % tag))

# ---- kernel statistics of the three profiled configs
with open(os.path.join(dst, '%s_bench_kernel_stats.md' % tag), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats summaries (round %s)\n\n' % tag[1:].lstrip('0'))
    f.write('command (per config, after one untraced run that writes the TunableOp GEMM choices): `rocprofv3 --kernel-trace --stats '
            '-d /tmp/prof_<cfg> -o bench -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 --config config/config_<cfg>.ini`, '
            'then `python tools/rocpd_stats.py /tmp/prof_<cfg>/bench_results.db --steps 7 --update` for the per-batch view '
            '(tools/prof_all.sh; sources under gpurun_out/ are scratch).  The default config\'s trace also contains the side '
            'measurements bench.py makes after the timed region (cacc_step_kernel<256,...> at E = 2^21, 60-launch graphs of the '
            'LSTM lock-step, 7 launches of the one-launch BPTT).\n\n')
    for cfg, title in CONFIGS:
        if not os.path.exists(os.path.join(src, '%s_kernel_stats_%s.csv' % (tag, cfg))):
            continue
        rows = list(csv.DictReader(open(os.path.join(src, '%s_kernel_stats_%s.csv' % (tag, cfg)))))
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        calls = sum(int(r['Calls']) for r in rows)
        f.write('## %s\n\ntotal kernel time %.2f ms over %d launches, %d distinct kernels\n\n' % (title, tot / 1e6, calls, len(rows)))
        f.write('| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n')
        for r in rows[:22]:
            f.write('| `%s` | %s | %.3f | %.2f | %.1f |\n' % (r['Name'][:100].replace('|', '/'), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                        float(r['AverageNs']) / 1e3, float(r['Percentage'])))
        txt = open(os.path.join(src, '%s_stats_%s.txt' % (tag, cfg))).read().splitlines()
        k = [i for i, l in enumerate(txt) if l.startswith('batch:')][0]
        f.write('\none batch in launch order -- update part (kernels > 30 us; loops summed), then the rollout:\n\n```\n')
        f.write('\n'.join(l[:150] for l in txt[k:]) + '\n```\n\n')

# ---- other configs
with open(os.path.join(dst, '%s_other_configs.md' % tag), 'w') as f:
    f.write('# All benchmarked configs (round %s)\n\n`python bench.py --no-cpu-baseline --steps 10 --config config/config_<cfg>.ini` on one '
            'MI355X (tools/prof_all.sh; the default row: `python bench.py --steps 20`); agent-steps/s = agents x replicas x lock-steps / s '
            'over rollout + update.  Round-1 numbers from profiles/r01r_other_configs.md.\n\n' % tag[1:].lstrip('0'))
    f.write('| config | agents x replicas | round 1 M/s | this round M/s | ms / batch | LSTM lock-step kernel (roofline.kernel) | us / launch | frac of fp32 MFMA peak |\n'
            '|---|---|---:|---:|---:|---|---:|---:|\n')
    d0 = last_json(os.path.join(src, '%s_bench_default.json' % tag))
    inline = {}                                   # round 4: bench.py's default line carries these three itself (`other_configs`)
    for o in d0.get('other_configs', []):
        for name in ('ma2c_nc_slowdown', 'ma2c_cnet_grid', 'ma2c_nc_catchup'):
            if ('config_%s.ini' % name) in o.get('workload', ''):
                inline[name] = o
    for name in ('default', 'ma2c_nc_slowdown', 'ma2c_cnet_grid', 'ma2c_nc_catchup', 'ma2c_cnet_catchup', 'ma2c_dial_catchup', 'ia2c_cu_catchup'):
        p = os.path.join(src, '%s_bench_%s.json' % (tag, name))
        na = 25 if 'grid' in name else 8
        if name in inline:
            o = inline[name]
            r = o.get('roofline', {})
            f.write('| %s | %d x %d | %s | **%.1f** | %.2f | %s | %.1f | %.2f |\n' % (
                LABEL[name] + ' [default line, other_configs]', na, 1024 if 'grid' in name else 4096, ROUND1.get(name, ''), o['value'] / 1e6,
                o['ms_per_step'], r.get('kernel', '').split(' (')[0], r.get('us_per_launch', 0.0), r.get('frac', 0.0)))
            rb = o.get('roofline_bptt', {})
            if 'frac' in rb:
                f.write('| | | | | | %s | %.1f | %.3f of HBM |\n' % (rb['kernel'].split(' (')[0], rb['us_per_launch'], rb['frac']))
            continue
        if not os.path.exists(p):
            continue
        d = last_json(p)
        r = d.get('roofline', {})
        f.write('| %s | %d x %d | %s | **%.1f** | %.2f | %s | %.1f | %.2f |\n' % (
            LABEL[name], na, d['config']['replicas_per_gpu'], ROUND1.get(name, ''), d['value'] / 1e6, d['ms_per_step'],
            r.get('kernel', '').split(' (')[0], r.get('us_per_launch', 0.0), r.get('frac', 0.0)))
    d = last_json(os.path.join(src, '%s_bench_default.json' % tag))
    f.write('\nSide measurements of the default run (bench.py JSON):\n\n')
    for k in ('roofline', 'roofline_bptt', 'roofline_env_step', 'roofline_env_step_large_E'):
        r = d.get(k, {})
        if 'frac' in r:
            f.write('* `%s`: %s -- %.1f us/launch, %.1f %s = %.3f of the %s roofline; PMC traffic %s\n' % (
                k, r['kernel'].split(' (')[0], r['us_per_launch'], r['achieved'], r['unit'], r['frac'], r['bound'],
                'n/a' if r.get('traffic') is None else '%.1f MB/launch (%s)' % (r['traffic'] / 1e6, r.get('traffic_source'))))
    cb = d.get('cpu_baseline')
    if cb:
        f.write('* `cpu_baseline`: %.0f %s on %d core (%s); all cores: %s; reference env only: %s\n' % (
            cb['value'], cb['unit'], cb['cores'], cb['kind'], json.dumps(cb.get('all_cores'))[:160], json.dumps(cb.get('reference_env_only'))[:200]))

# ---- verbatim copies
for a, b in (('%s_pmc_traffic.json', '%s_pmc_traffic.json'), ('%s_pmc_traffic.md', '%s_pmc_traffic.md'), ('%s_pmc_lstm.md', '%s_pmc_lstm_stalls.md'),
             ('%s_time_fused.log', '%s_time_fused.txt'), ('%s_env_microbench.log', '%s_env_microbench.txt'),
             ('%s_step_timeline.txt', '%s_step_timeline.txt'), ('%s_bptt_timeline.txt', '%s_bptt_timeline.txt'),
             ('%s_mfma_valu_overlap.txt', '%s_mfma_valu_overlap.txt'), ('%s_train_speed.txt', '%s_train_speed.txt'),
             ('%s_learn_grid.json', '%s_learn_grid.json'), ('%s_untraced_breakdown.md', '%s_untraced_breakdown.md'),
             ('%s_clock_power.txt', '%s_clock_power.txt'), ('%s_ab_lockstep.txt', '%s_ab_lockstep.txt'), ('%s_fc_pair.txt', '%s_fc_pair.txt'),
             ('%s_grid_step.txt', '%s_grid_step.txt'), ('%s_determinism.txt', '%s_determinism.txt'),
             ('%s_ab_lockstep_nc.txt', '%s_ab_lockstep_nc_pass2.txt'), ('%s_e1_path.txt', '%s_e1_path.txt'),
             ('%s_ab_round6_switches.txt', '%s_ab_round6_switches.txt'), ('%s_heads_loss.txt', '%s_heads_loss.txt'),
             ('%s_graph_nodes.txt', '%s_graph_nodes.txt'), ('%s_pmc_sq.md', '%s_pmc_sq.md')):
    if os.path.exists(os.path.join(src, a % tag)):
        shutil.copy(os.path.join(src, a % tag), os.path.join(dst, b % tag))
print('profiles/%s_* written' % tag)
