#!/bin/bash
# usage: tools/kstat_graph.sh <tag> [bench args...]  -- rocprofv3 kernel trace of a hipGraph-replayed bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && export JT_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$tag -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/ks_$tag/k_kernel_stats.csv 40 30
