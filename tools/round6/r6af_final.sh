#!/bin/bash
# round 6, run af: the driver's command on the round's final code (calling-thread backward, planned Adam step, pose-only march with
# stored derivatives, lattice kernel) + the test-time optimisation trace
cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/round6_bench_full_line.json 2> gpurun_out/round6_bench_full.err
tail -4 gpurun_out/round6_bench_full.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6af_trace -o k -- python3 $GRAFT_REPO_ROOT/tools/eval_bench.py --no-render --graph --test-iters 100 > $GRAFT_REPO_ROOT/gpurun_out/r6af.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py gpurun_out/r6af_trace/k_kernel_stats.csv 40 200 > gpurun_out/round6_testoptim_trace_summary.txt
rm -rf gpurun_out/r6af_trace
head -12 gpurun_out/round6_testoptim_trace_summary.txt
