#!/bin/bash
# evidence for the round's final code: default (headline) and LLFF final grid -- bench line, kernel trace + stats, the two --pmc
# passes, launches per iteration; the tile-owned variant's trace; the test-time optimisation trace
cd $GRAFT_REPO_ROOT
bash tools/profile_cmd.sh round5_default > gpurun_out/round5_default_profile.log 2>&1
bash tools/profile_cmd.sh round5_llff --config bat_llff_VM_MLP > gpurun_out/round5_llff_profile.log 2>&1
NO_PMC=1 JT_BWD_SPLIT=1 JT_TILE_CFG=1 bash tools/profile_cmd.sh round5_tile > gpurun_out/round5_tile_profile.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/round5_testoptim -o k -- python3 $GRAFT_REPO_ROOT/tools/eval_bench.py --no-render --graph --test-iters 100 > $GRAFT_REPO_ROOT/gpurun_out/round5_testoptim.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py gpurun_out/round5_testoptim/k_kernel_stats.csv 14 200 > gpurun_out/round5_testoptim_trace_summary.txt
rm -f gpurun_out/round5_testoptim/*kernel_trace.csv
head -12 gpurun_out/round5_default_trace_summary.txt; cat gpurun_out/round5_default_launches_per_iteration.json; head -14 gpurun_out/round5_llff_trace_summary.txt; head -12 gpurun_out/round5_tile_trace_summary.txt; head -10 gpurun_out/round5_testoptim_trace_summary.txt
