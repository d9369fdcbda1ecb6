#!/bin/bash
# Round-end evidence on the GPU box: kernel stats of the default bench, the probe, and the two --pmc passes
# (separate runs, --kernel-trace only, as gpurun requires).  Results under gpurun_out/rp_*.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rp_stage4 -o k -- python $R/bench.py --stage 4 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/rp_stage4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rp_probe -o k -- python $R/bench.py --probe-only > $R/gpurun_out/rp_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/rp_pmc_FETCH_SIZE -o p -- python $R/bench.py --probe-only > $R/gpurun_out/rp_pmc_FETCH_SIZE.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/rp_pmc_WRITE_SIZE -o p -- python $R/bench.py --probe-only > $R/gpurun_out/rp_pmc_WRITE_SIZE.log 2>&1
cd $R
python bench.py --probe-only 2>/dev/null | tail -1 > gpurun_out/rp_probe.json
python tools/prof_summary.py gpurun_out/rp_stage4/k_kernel_stats.csv 30 25 > gpurun_out/rp_stage4_summary.txt
python tools/prof_summary.py gpurun_out/rp_probe/k_kernel_stats.csv 12 > gpurun_out/rp_probe_summary.txt
python tools/pmc_summary.py gpurun_out/rp_pmc_FETCH_SIZE/p_counter_collection.csv k_shade > gpurun_out/rp_pmc_FETCH_SIZE_summary.txt
python tools/pmc_summary.py gpurun_out/rp_pmc_WRITE_SIZE/p_counter_collection.csv k_shade > gpurun_out/rp_pmc_WRITE_SIZE_summary.txt
python tools/pmc_traffic.py gpurun_out/rp_pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/rp_pmc_WRITE_SIZE/p_counter_collection.csv gpurun_out/rp_probe.json gpurun_out/rp_pmc_traffic.json > /dev/null
