#!/bin/bash
# Round-2 evidence on the GPU box: the default bench command (secondary legs off) under (1) the kernel trace, (2)/(3) two
# separate --pmc passes (gpurun requires --pmc runs to carry --kernel-trace only).  Results under gpurun_out/r2_*.
R=$GRAFT_REPO_ROOT
CMD="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
python3 $CMD > $R/gpurun_out/r2_bench_line.json 2> $R/gpurun_out/r2_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_trace -o k -- python3 $CMD > $R/gpurun_out/r2_trace.log 2>&1
JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_trace_1s -o k -- python3 $CMD > $R/gpurun_out/r2_trace_1s.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_FETCH_SIZE -o p -- python3 $CMD > $R/gpurun_out/r2_pmc_FETCH_SIZE.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_WRITE_SIZE -o p -- python3 $CMD > $R/gpurun_out/r2_pmc_WRITE_SIZE.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/r2_trace/k_kernel_stats.csv 30 27 > gpurun_out/r2_trace_summary.txt
python3 tools/prof_summary.py gpurun_out/r2_trace_1s/k_kernel_stats.csv 30 27 > gpurun_out/r2_trace_1s_summary.txt
python3 tools/pmc_summary.py gpurun_out/r2_pmc_FETCH_SIZE/p_counter_collection.csv k_shade k_march k_wgrad > gpurun_out/r2_pmc_FETCH_SIZE_summary.txt
python3 tools/pmc_summary.py gpurun_out/r2_pmc_WRITE_SIZE/p_counter_collection.csv k_shade k_march k_wgrad > gpurun_out/r2_pmc_WRITE_SIZE_summary.txt
python3 tools/pmc_traffic_instep.py gpurun_out/r2_pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/r2_pmc_WRITE_SIZE/p_counter_collection.csv gpurun_out/r2_bench_line.json gpurun_out/r2_pmc_traffic_instep.json > /dev/null
grep -E '"roofline"' -o gpurun_out/r2_bench_line.json | head -1
cat gpurun_out/r2_trace_summary.txt | head -14
cat gpurun_out/r2_pmc_traffic_instep.json | head -30
