#!/bin/bash
# Round-4 evidence set on the GPU box (copy the summaries into profiles/round4_*): bench line + kernel trace + FETCH / WRITE passes
# of the default command and of the LLFF final grid (tools/profile_cmd.sh), the SQ / TA passes and pipe-busy fractions of the default.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
bash tools/profile_cmd.sh r4_default > gpurun_out/r4_default_profile.log 2>&1
bash tools/profile_cmd.sh r4_llff --config bat_llff_VM_MLP > gpurun_out/r4_llff_profile.log 2>&1
bash tools/pmc_sq3.sh r4 > gpurun_out/r4_sq3.txt 2>&1
bash tools/pmc_ta.sh r4 > gpurun_out/r4_ta.txt 2>&1
python3 tools/pipe_busy.py gpurun_out/sq3_r4/k_counter_collection.csv gpurun_out/ta_r4/k_counter_collection.csv gpurun_out/r4_default_trace/k_kernel_stats.csv 27 gpurun_out/r4_default_pipe_busy.json
head -20 gpurun_out/r4_default_trace_summary.txt | cut -c1-140
head -14 gpurun_out/r4_llff_trace_summary.txt | cut -c1-140
rm -f gpurun_out/sq3_r4/*kernel_trace.csv gpurun_out/ta_r4/*kernel_trace.csv
