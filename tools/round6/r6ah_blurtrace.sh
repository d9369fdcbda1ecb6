#!/bin/bash
# round 6, run ah: kernel trace of the blurred stage-4 iteration (it 9 000)
cd $GRAFT_REPO_ROOT
NO_PMC=1 bash tools/profile_cmd.sh r6ah_blur --it 9000 > gpurun_out/r6ah_profile.log 2>&1
head -40 gpurun_out/r6ah_blur_trace_summary.txt
cat gpurun_out/r6ah_blur_bench_line.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): j = json.loads(l); print(j['ms_per_step'], j['config']['workload'])"
