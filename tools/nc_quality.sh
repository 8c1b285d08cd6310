#!/bin/bash
# the full NeurComm slow-down schedule (16 667 updates = 1e6 lock-steps per replica) through
# the SAME batched product at E = 8 / 64 / 512 (two seeds each) and E = 4096 (second seed; seed 12 is profiles/r02_learn_*),
# all processes concurrently on the one GPU (the small-E runs are launch-bound).  Output: gpurun_out/learn_*.json
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
pids=""
for E in 8 64 512; do
  for s in 12 13; do
    python tools/learn_curve.py ma2c_nc slowdown $E 16667 500 $s > gpurun_out/nc_quality_E${E}_s${s}.log 2>&1 &
    pids="$pids $!"
  done
done
python tools/learn_curve.py ma2c_nc slowdown 4096 16667 500 13 > gpurun_out/nc_quality_E4096_s13.log 2>&1 &
pids="$pids $!"
for p in $pids; do wait $p; done
tail -n 2 gpurun_out/nc_quality_E*.log
