#!/bin/bash
# the driver's command on the round's final code
cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/round6_bench_full_line.json 2> gpurun_out/round6_bench_full.err
tail -4 gpurun_out/round6_bench_full.err
