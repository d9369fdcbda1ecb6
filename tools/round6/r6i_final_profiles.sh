#!/bin/bash
# round 6: evidence for the round's code -- the driver's command (full line with extras), then kernel trace + stats + the two
# --pmc passes of the default and of the LLFF final grid, and the default with every kernel alone
cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/round6_bench_full_line.json 2> gpurun_out/round6_bench_full.err
bash tools/profile_cmd.sh round6_default > gpurun_out/round6_default_profile.log 2>&1
bash tools/profile_cmd.sh round6_llff --config bat_llff_VM_MLP > gpurun_out/round6_llff_profile.log 2>&1
JT_NO_AUX=1 JT_ADAM_EARLY=0 JT_SCATTER_WGS=256 NO_PMC=1 bash tools/profile_cmd.sh round6_default_noaux > gpurun_out/round6_default_noaux_profile.log 2>&1
JT_LEAN_TAPE=0 NO_PMC=1 bash tools/profile_cmd.sh round6_default_fulltape > gpurun_out/round6_default_fulltape_profile.log 2>&1
head -20 gpurun_out/round6_default_trace_summary.txt; head -14 gpurun_out/round6_default_noaux_trace_summary.txt; tail -3 gpurun_out/round6_bench_full.err
