#!/bin/bash
# usage: tools/kstat.sh <tag> [bench args...]  -- rocprofv3 kernel stats of a short bench run (single stream, headline leg only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && export JT_NO_AUX=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$tag -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py gpurun_out/ks_$tag/k_kernel_stats.csv 16 27
