export TMPDIR=/tmp
python bench.py --steps 20 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
for c in ia2c_fp_catchup ma2c_nc_slowdown ma2c_cnet_grid; do
  python bench.py --no-cpu-baseline --steps 2 --warmup 2 --config config/config_$c.ini > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o bench -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 --config config/config_$c.ini > gpurun_out/r02_prof_$c.log 2>&1
  python tools/rocpd_stats.py /tmp/prof_$c/bench_results.db --steps 7 --update > gpurun_out/r02_stats_$c.txt 2>&1
  cp /tmp/prof_$c/bench_kernel_stats.csv gpurun_out/r02_kernel_stats_$c.csv
done
for c in ma2c_nc_slowdown ma2c_cnet_grid ma2c_nc_catchup ma2c_dial_catchup ma2c_cnet_catchup ia2c_cu_catchup; do
  python bench.py --no-cpu-baseline --steps 10 --config config/config_$c.ini > gpurun_out/r02_bench_$c.json 2> gpurun_out/r02_bench_$c.err
done
mkdir -p /tmp/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o pmc_FETCH_SIZE --output-format csv -- python tools/pmc_env.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o pmc_WRITE_SIZE --output-format csv -- python tools/pmc_env.py > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc r02 gpurun_out > gpurun_out/r02_pmc_summary.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace -d /tmp/pmc_a -o a --output-format csv -- python tools/time_fused.py > gpurun_out/r02_time_fused.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_b -o b --output-format csv -- python tools/time_fused.py > /dev/null 2>&1
mkdir -p /tmp/pmc_ab; cp /tmp/pmc_a/*counter_collection.csv /tmp/pmc_ab/a_counter_collection.csv; cp /tmp/pmc_b/*counter_collection.csv /tmp/pmc_ab/b_counter_collection.csv
python tools/pmc_table.py /tmp/pmc_ab lstm_step lstm_bptt > gpurun_out/r02_pmc_lstm.md 2>&1
python tools/time_fused.py > gpurun_out/r02_time_fused.log 2>&1
python tools/env_microbench.py > gpurun_out/r02_env_microbench.log 2>&1
python -c "
import json
d=json.loads(open('gpurun_out/r02_bench_default.json').read())
print(d['value']/1e6, d['ms_per_step'])
for k in d:
    if k.startswith('roofline'): print(k, d[k].get('frac'), d[k].get('us_per_launch'), d[k].get('traffic'))
print(json.dumps(d.get('cpu_baseline'))[:600])
"
