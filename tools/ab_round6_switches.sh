#!/bin/bash
# Same-box A/B of the round-6 forms that have an environment switch (ONE library, alternating runs):
#   NMARL_GRID_ENV_IN_KERNEL  the grid env step as a role of CommNet's lock-step launch (csrc/lstm_mfma.hip GENV)  vs  nmarl_grid_step
#   NMARL_MSG_CARRY           the re-step's message term handed to the next lock-step (CARRY 1 | 2)               vs  recomputed
#   NMARL_FUSED_HEADS_LOSS    heads + loss + heads' backward in one pass (nmarl_heads_loss)                       vs  GEMM + loss fwd / bwd + thin_bwd
#   NMARL_BPTT_HEAD_DY        the one-launch BPTT expands dy8 itself (nmarl_lstm_bptt_seq_dy)                      vs  dL/dh as a tensor
#   NMARL_BPTT_HEAD_DY_COUPLED  the same for the coupled BPTT kernels (nmarl_bptt_coupled_t.dy8)                    vs  dL/dh as a tensor
cd "$(dirname "$0")/.."
run() { env "${@:3}" python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 4 --config config/config_$1.ini 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; u=d.get('update') or {}
print('%-28s %-18s %7.3f ms %6.1f M | lock-step %6.2f us | rollout %7.1f us | update %s us' % ('$2', '$1', d['ms_per_step'], d['value']/1e6, r.get('us_per_launch',0), r.get('rollout_graph_us') or 0, ('%.1f' % u['update_graph_us']) if u.get('update_graph_us') else 'n/a'))"; }
for i in 1 2 3; do
  run ma2c_cnet_grid   "grid: env kernel"        NMARL_GRID_ENV_IN_KERNEL=0
  run ma2c_cnet_grid   "grid: env role (default)" NMARL_GRID_ENV_IN_KERNEL=1
  run ma2c_cnet_grid   "grid: no carry"          NMARL_MSG_CARRY=0
  run ma2c_nc_slowdown "NeurComm: no carry"      NMARL_MSG_CARRY=0
  run ma2c_nc_slowdown "NeurComm: default"       NMARL_MSG_CARRY=1
  run ia2c_fp_catchup  "IA2C-FP: autograd chain" NMARL_FUSED_HEADS_LOSS=0
  run ia2c_fp_catchup  "IA2C-FP: fused, dh tensor" NMARL_BPTT_HEAD_DY=0
  run ia2c_fp_catchup  "IA2C-FP: default"        NMARL_BPTT_HEAD_DY=1
  run ma2c_nc_slowdown "NeurComm: fused, dh tensor" NMARL_BPTT_HEAD_DY_COUPLED=0
  run ma2c_cnet_grid   "grid: fused, dh tensor"  NMARL_BPTT_HEAD_DY_COUPLED=0
  run ma2c_nc_slowdown "NeurComm: autograd chain" NMARL_FUSED_HEADS_LOSS=0
  run ma2c_cnet_grid   "grid: autograd chain"    NMARL_FUSED_HEADS_LOSS=0
done
