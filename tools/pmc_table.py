"""Average per-launch PMC counter values per kernel from rocprofv3 `--pmc ... --output-format csv` runs:
    python tools/pmc_table.py <dir> [kernel-name-substring ...]  ->  markdown table on stdout
(every *_counter_collection.csv below <dir>; the first 2 launches of each kernel dropped as warm-up)."""
import collections
import csv
import glob
import os
import sys

d, subs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if subs and not any(s in k for s in subs):
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[(k, r['Counter_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('| kernel | counter | launches | average per launch | avg duration us (profiled) |\n|---|---|---:|---:|---:|')
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c][2:] or acc[k][c]
        t = dur[(k, c)][2:] or dur[(k, c)]
        print('| `%s` | %s | %d | %.4g | %.1f |' % (k[:70], c, len(v), sum(v) / len(v), sum(t) / len(t) / 1e3))
