#!/bin/bash
# Instruction-fetch counters of the matrix-core lock-step kernels (is straight-line prologue / epilogue code fetch-bound?):
# one rocprofv3 --pmc pass over tools/pmc_env.py at a small env size.  -> gpurun_out/<tag>_pmc_icache.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PMC_E=65536 PMC_EG=8192
TAG=${1:-r04}
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INSTS_SALU\|SQ_INSTS_VALU\b" | sort -u > /tmp/avail.txt
echo "available:" $(cat /tmp/avail.txt)
rm -rf /tmp/pmci; mkdir -p /tmp/pmci
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_IFETCH_LEVEL SQ_BUSY_CYCLES"; do
  n=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmci -o p_$n --output-format csv -- python tools/pmc_env.py > /tmp/pmci/log_$n.txt 2>&1
done
python - <<'PY' > gpurun_out/${TAG}_pmc_icache.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmci/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'lstm_step_x' in k or 'lstm_bptt' in k or 'dial' in k:
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = v[2:] or v
        print('   %-22s launches %3d  avg %.4g' % (c, len(v), sum(v) / len(v)))
PY
cat gpurun_out/${TAG}_pmc_icache.txt
