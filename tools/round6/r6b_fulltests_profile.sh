#!/bin/bash
# round 6, run b: the whole GPU suite on the lean-tape default, then the evidence set of the default bench command
cd $GRAFT_REPO_ROOT
( time timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r6b_fulltests.log 2>&1
tail -n 30 gpurun_out/r6b_fulltests.log
bash tools/profile_cmd.sh round6_default > gpurun_out/round6_default_profile.log 2>&1
JT_NO_AUX=1 JT_ADAM_EARLY=0 NO_PMC=1 bash tools/profile_cmd.sh round6_default_noaux > gpurun_out/round6_default_noaux_profile.log 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
JT_GRAPH=1 $B > gpurun_out/r6b_bench_graph.json 2>/dev/null
$B > gpurun_out/r6b_bench_eager.json 2>/dev/null
head -24 gpurun_out/round6_default_trace_summary.txt
