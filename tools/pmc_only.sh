cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; TAG=r04; mkdir -p /tmp/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o pmc_FETCH_SIZE --output-format csv -- python tools/pmc_env.py > gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o pmc_WRITE_SIZE --output-format csv -- python tools/pmc_env.py > gpurun_out/${TAG}_pmc_write.log 2>&1
python tools/pmc_summary.py /tmp/pmc $TAG gpurun_out > gpurun_out/${TAG}_pmc_summary.log 2>&1
tail -3 gpurun_out/${TAG}_pmc_fetch.log; cat gpurun_out/${TAG}_pmc_traffic.md
