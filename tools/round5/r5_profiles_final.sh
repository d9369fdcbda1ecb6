#!/bin/bash
# evidence for the round's LAST code (scatter on 192 CUs beside the forked GEMMs, early optimizer step): default (headline) and LLFF
# final grid -- bench line, kernel trace + stats, the two --pmc passes, launches per iteration
cd $GRAFT_REPO_ROOT
bash tools/profile_cmd.sh round5_default > gpurun_out/round5_default_profile.log 2>&1
bash tools/profile_cmd.sh round5_llff --config bat_llff_VM_MLP > gpurun_out/round5_llff_profile.log 2>&1
JT_NO_AUX=1 JT_ADAM_EARLY=0 NO_PMC=1 bash tools/profile_cmd.sh round5_default_noaux > gpurun_out/round5_default_noaux_profile.log 2>&1
head -14 gpurun_out/round5_default_trace_summary.txt; cat gpurun_out/round5_default_launches_per_iteration.json; cat gpurun_out/round5_default_pmc_traffic.json; head -12 gpurun_out/round5_llff_trace_summary.txt; head -12 gpurun_out/round5_default_noaux_trace_summary.txt
