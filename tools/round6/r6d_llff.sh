#!/bin/bash
# round 6, run d: why did the LLFF final grid get slower with the lean tape in run c (2.31 against 2.12 ms)?  traces of both
cd $GRAFT_REPO_ROOT
NO_PMC=1 bash tools/profile_cmd.sh r6d_llff_lean --config bat_llff_VM_MLP > gpurun_out/r6d_llff_lean.log 2>&1
JT_LEAN_TAPE=0 NO_PMC=1 bash tools/profile_cmd.sh r6d_llff_full --config bat_llff_VM_MLP > gpurun_out/r6d_llff_full.log 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc --config bat_llff_VM_MLP"
for i in 1 2; do
$B > gpurun_out/r6d_bench_llff_lean_$i.json 2>/dev/null
JT_LEAN_TAPE=0 $B > gpurun_out/r6d_bench_llff_full_$i.json 2>/dev/null
JT_NO_AUX=1 $B > gpurun_out/r6d_bench_llff_lean_noaux_$i.json 2>/dev/null
done
timeout 600 python -m pytest tests/test_engine_trace.py -x -q -s 2>&1 | tail -12 > gpurun_out/r6d_engine.txt
head -14 gpurun_out/r6d_llff_lean_trace_summary.txt; head -14 gpurun_out/r6d_llff_full_trace_summary.txt; cat gpurun_out/r6d_engine.txt
