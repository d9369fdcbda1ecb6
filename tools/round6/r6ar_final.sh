#!/bin/bash
# round 6, run ar: evidence on the round's final code: the driver's command, the LLFF profiler passes, the whole GPU suite + smoke
cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/round6_bench_full_line.json 2> gpurun_out/round6_bench_full.err
tail -4 gpurun_out/round6_bench_full.err
bash tools/profile_cmd.sh round6_llff --config bat_llff_VM_MLP > gpurun_out/round6_llff_profile.log 2>&1
head -14 gpurun_out/round6_llff_trace_summary.txt
bash tools/round6/r6m_fulltests.sh 2>&1 | tail -8
