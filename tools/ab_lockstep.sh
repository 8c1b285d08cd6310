#!/bin/bash
# A/B of the three forms of the IA2C-FP lock-step on ONE box (round 5): (a) two launches -- lstm_step_x_kernel<3,0,0> + cacc_step_encode_kernel
# (round 4), (b) the input encoders inside the lock-step kernel + the env kernel, (c) one launch (encoders + env step inside).  Interleaved,
# two passes, headline bench without side configs.   bash tools/ab_lockstep.sh > profiles/rNN_ab_lockstep.txt
cd "$(dirname "$0")/.."
echo "# python bench.py --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline, same box, interleaved (NMARL_INKERNEL_ENCODE / NMARL_INKERNEL_ENV / NMARL_FC_BWD_PAIR; the last form: the update's encoder backward as two launches reading S)"
echo "# form | ms per batch | M env-steps/s | lock-step launch us (in rollout, by graph difference) | rollout graph us | rollout graph without the lock-step launches us | update graph us"
for pass in 1 2; do
  for form in "0 0 1 two-launches(r4)" "1 0 1 encoders-inside" "1 1 1 one-launch" "1 1 0 one-launch,two-fc_bwd-launches"; do
    set -- $form
    NMARL_INKERNEL_ENCODE=$1 NMARL_INKERNEL_ENV=$2 NMARL_FC_BWD_PAIR=$3 timeout 300 python bench.py --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$4 | %.3f | %.1f | %.2f | %.1f | %.1f | %.1f' % (d['ms_per_step'], d['value']/1e6, r['us_per_launch'], r['rollout_graph_us'], r['rollout_graph_us_without_lstm_steps'], d.get('update',{}).get('update_graph_us', float('nan'))))"
  done
done
