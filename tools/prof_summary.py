"""Condense a rocprofv3 --kernel-trace --stats run (…_kernel_stats.csv) into a small tracked summary
under profiles/.   python tools/prof_summary.py gpurun_out/prof_r1/bench_kernel_stats.csv profiles/r01_bench_kernel_stats.md "<command>" """
import csv
import sys

src, dst, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else '')
rows = list(csv.DictReader(open(src)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(dst, 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats summary\n\ncommand: `%s`\n\nsource: %s (scratch, not tracked)\n\n' % (cmd, src))
    f.write('total kernel time: %.2f ms over %d distinct kernels\n\n' % (tot / 1e6, len(rows)))
    f.write('| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n')
    for r in rows[:40]:
        f.write('| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %.1f |\n' % (
            r['Name'][:110].replace('|', '/'), r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
            float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))
print('wrote', dst)
