#!/bin/bash
# rocprofv3 kernel trace of bench.py for the given configs: tools/prof_cfg.sh <tag> <cfg> [<cfg> ...] -> gpurun_out/<tag>_stats_<cfg>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
for c in "$@"; do
  python bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 2 --config config/config_$c.ini > /dev/null 2>&1      # writes the TunableOp choices
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o bench -- python bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 --config config/config_$c.ini > gpurun_out/${tag}_prof_$c.log 2>&1
  python tools/rocpd_stats.py /tmp/prof_$c/bench_results.db --steps 7 --update > gpurun_out/${tag}_stats_$c.txt 2>&1
  cp /tmp/prof_$c/bench_kernel_stats.csv gpurun_out/${tag}_kernel_stats_$c.csv 2>/dev/null
done
