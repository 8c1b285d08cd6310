#!/bin/bash
# A/B of NeurComm's lock-step forms on ONE box (round 6), BASELINE configs[2] (slow-down) and configs[4] per GPU (catch-up), 8 x 4096:
# (a) two launches per lock-step -- lstm_step_x_kernel<4,1,0> + cacc_step_encode_kernel (round 5), (b) the input encoders inside the
# lock-step kernel (<4,1,1>) + the env kernel, (c) ONE launch (encoders + env step inside).  Interleaved, two passes.
#   bash tools/ab_lockstep_nc.sh > profiles/rNN_ab_lockstep_nc.txt
cd "$(dirname "$0")/.."
echo "# python bench.py --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline --config config/config_ma2c_nc_<scenario>.ini, same box, interleaved (NMARL_NC_ONE_LAUNCH / NMARL_NC_ENV_IN_KERNEL)"
echo "# scenario | form | ms per batch | M env-steps/s | lock-step launch us (in rollout, by graph difference) | rollout graph us | rollout graph without the lock-step launches us | update graph us"
for pass in 1 2; do
  for sc in slowdown catchup; do
    for form in "0 0 two-launches(r5)" "1 0 encoders-inside" "1 1 one-launch"; do
      set -- $form
      NMARL_NC_ONE_LAUNCH=$1 NMARL_NC_ENV_IN_KERNEL=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline --config config/config_ma2c_nc_$sc.ini 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$sc | $3 | %.3f | %.1f | %.2f | %.1f | %.1f | %.1f' % (d['ms_per_step'], d['value']/1e6, r.get('us_per_launch', float('nan')), r.get('rollout_graph_us') or float('nan'), r.get('rollout_graph_us_without_lstm_steps') or float('nan'), (d.get('update') or {}).get('update_graph_us') or float('nan')))"
    done
  done
done
