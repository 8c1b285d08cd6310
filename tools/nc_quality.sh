#!/bin/bash
# the full NeurComm slow-down schedule (16 667 updates = 1e6 lock-steps per replica) through
# the SAME batched product at E = 8 / 64 / 512 and a second seed at E = 4096 (seed 12: profiles/r02_learn_*), one run after
# the other (seven concurrent processes on one GPU ran 10x slower each and were cut by the time limit: gpurun_out of the
# first attempt is kept under profiles/r03_nc_quality_partial_*).  Rows are flushed as they come (python -u).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "8 12" "64 12" "512 12" "4096 13"; do
  set -- $spec
  timeout ${NC_QUALITY_TIMEOUT:-400} python -u tools/learn_curve.py ma2c_nc slowdown $1 16667 500 $2 > gpurun_out/nc_quality_E$1_s$2.log 2>&1
  tail -n 1 gpurun_out/nc_quality_E$1_s$2.log | cut -c1-300
done
