#!/bin/bash
# Measurement pass on one MI355X (rounds 5-6): default bench line + untraced breakdown + clock / power samples + same-box A/B of the lock-step forms,
# rocprofv3 kernel traces of the BASELINE configs (part 1: one box for bench line and traces); the other configs, PMC traffic passes (separate FETCH / WRITE runs),
# timelines, side benches (part 2).  Results under gpurun_out/ (python tools/make_profiles.py rNN copies the summaries into profiles/).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r06}
PART=${2:-all}            # 1: bench line + traces (one box for r05_bench_default.json and r05_bench_kernel_stats.md), 2: counters / timelines / side benches
if [ "$PART" != 2 ]; then
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python tools/untraced_breakdown.py gpurun_out/${TAG}_bench_default.json ${TAG} > gpurun_out/${TAG}_untraced_breakdown.md 2>&1
python tools/clock_power.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_clock_power.txt
bash tools/ab_lockstep.sh > gpurun_out/${TAG}_ab_lockstep.txt 2>&1
bash tools/ab_lockstep_nc.sh > gpurun_out/${TAG}_ab_lockstep_nc.txt 2>&1
python tools/e1_path.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_e1_path.txt
bash tools/ab_round6_switches.sh > gpurun_out/${TAG}_ab_round6_switches.txt 2>&1
( python tools/time_heads_loss.py; python tools/time_heads_loss.py 25 1024 120 5 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_heads_loss.txt
( echo "# Captured-graph node census: python tools/graph_nodes.py <env> <agent> <E> [--phases] -- node types read back through hipGraphGetNodes / hipGraphNodeGetType"
  echo "# (round 5's tree: rollout graphs held 1 (ia2c_fp) / 11 (ma2c_nc) / 10 (grid) memcpy nodes, update graphs 1-2)"
  python tools/graph_nodes.py cacc ia2c_fp 4096; python tools/graph_nodes.py grid ma2c_ic3 1024; python tools/graph_nodes.py cacc ma2c_nc 4096 --phases ) 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v run_backward > gpurun_out/${TAG}_graph_nodes.txt
bash tools/prof_cfg.sh $TAG ia2c_fp_catchup ma2c_nc_slowdown ma2c_cnet_grid ma2c_dial_catchup
fi
if [ "$PART" = 1 ]; then exit 0; fi
for c in ma2c_dial_catchup ma2c_cnet_catchup ia2c_cu_catchup; do      # (NeurComm slow-down / catch-up and the CommNet grid: `other_configs` of the default line)
  python bench.py --no-cpu-baseline --steps 10 --config config/config_$c.ini > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
done
mkdir -p /tmp/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o pmc_FETCH_SIZE --output-format csv -- python tools/pmc_env.py > gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o pmc_WRITE_SIZE --output-format csv -- python tools/pmc_env.py > gpurun_out/${TAG}_pmc_write.log 2>&1
python tools/pmc_summary.py /tmp/pmc $TAG gpurun_out > gpurun_out/${TAG}_pmc_summary.log 2>&1
# SQ counters (issue / stall / matrix-pipe busy, LDS conflicts) of the same harness: two more passes, counters only
mkdir -p /tmp/pmcsq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d /tmp/pmcsq -o sqa --output-format csv -- python tools/pmc_env.py > gpurun_out/${TAG}_pmc_sqa.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d /tmp/pmcsq -o sqb --output-format csv -- python tools/pmc_env.py > gpurun_out/${TAG}_pmc_sqb.log 2>&1
python tools/pmc_sq.py /tmp/pmcsq > gpurun_out/${TAG}_pmc_sq.md 2>&1
# shader-clock phase timelines of the lock-step kernels (instrumentation build of csrc/lstm_mfma.hip, tools/step_timeline.py)
python tools/time_fc_pair.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_fc_pair.txt
python tools/time_grid_step.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_grid_step.txt
python tools/determinism.py ma2c_nc slowdown 1000 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v run_backward > gpurun_out/${TAG}_determinism.txt
python tools/determinism.py ia2c_fp catchup 500 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v run_backward >> gpurun_out/${TAG}_determinism.txt
python tools/determinism.py ma2c_ic3 grid 1500 1024 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v run_backward >> gpurun_out/${TAG}_determinism.txt
python tools/microbench/mfma_valu_overlap.py > gpurun_out/${TAG}_mfma_valu_overlap.txt 2>&1
python tools/train_speed.py > gpurun_out/${TAG}_train_speed.txt 2>&1
python tools/train_speed.py config/config_ma2c_nc_slowdown.ini 200 >> gpurun_out/${TAG}_train_speed.txt 2>&1
python tools/bptt_timeline.py --build > /dev/null 2>&1
( echo "## python tools/bptt_timeline.py nc  (NeurComm: 8 x 4096 rows, T = 60)"; python tools/bptt_timeline.py nc 2>&1 | grep -v amdgpu.ids; echo
  echo "## python tools/bptt_timeline.py grid  (CommNet grid: 25 x 1024 rows, T = 120)"; python tools/bptt_timeline.py grid 2>&1 | grep -v amdgpu.ids; echo
  echo "## python tools/bptt_timeline.py seq  (uncoupled one-launch BPTT, 8 x 4096 rows, T = 60: the clock this box sustains)"; python tools/bptt_timeline.py seq 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${TAG}_bptt_timeline.txt
python tools/step_timeline.py --build > /dev/null 2>&1
( echo "## python tools/step_timeline.py 4  (NeurComm shape: 8 x 4096 rows, KX = 192, one-launch policy + value step)"
  python tools/step_timeline.py 4 2>&1 | grep -v amdgpu.ids; echo
  echo "## python tools/step_timeline.py 4 grid  (CommNet grid shape: 25 x 1024 rows, KX = 64, 4 neighbours, encoder inside; block 0 = a corner agent)"
  python tools/step_timeline.py 4 grid 2>&1 | grep -v amdgpu.ids; echo
  echo "## python tools/step_timeline.py 3  (IA2C-FP shape, policy + value step: lstm_step_x_kernel<3,0,0>, x read from HBM)"
  python tools/step_timeline.py 3 2>&1 | grep -v amdgpu.ids; echo
  echo "## python tools/step_timeline.py 3 enc  (lstm_step_x_kernel<3,0,1>: the input encoders in the pre-phase; no env step in this harness)"
  python tools/step_timeline.py 3 enc 2>&1 | grep -v amdgpu.ids; echo
  echo "## python tools/step_timeline.py 3 enc noout  (the same without the saved-activation stores: the bootstrap step)"
  python tools/step_timeline.py 3 enc noout 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${TAG}_step_timeline.txt
if [ "$PART" = 2 ]; then exit 0; fi
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_bench_default.json').read().strip().splitlines()[-1])
print(d['value']/1e6, d['ms_per_step'])
for k in d:
    if k.startswith('roofline') and isinstance(d[k], dict): print(k, d[k].get('frac'), d[k].get('us_per_launch'), d[k].get('traffic'), d[k].get('error'))
print(json.dumps(d.get('cpu_baseline'))[:700])
"
