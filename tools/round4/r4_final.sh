#!/bin/bash
# end-of-round evidence: whole GPU suite + smoke, the profile set, the full bench line, the 40 000-iteration converge run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=gpurun_out
bash tools/round4/r4_fulltests.sh
bash tools/round4_profiles.sh > $O/r4_final_profiles.log 2>&1; tail -3 $O/r4_final_profiles.log | cut -c1-140
python3 bench.py > $O/r4_final_bench_full.json 2> $O/r4_final_bench_full.err; python3 -c "
import json; j=json.loads([l for l in open('$O/r4_final_bench_full.json') if l.startswith('{')][-1]); print('bench', j['value'], j['ms_per_step'], j['vs_baseline'], j['roofline']['frac'], j['roofline'].get('traffic_source'), j['roofline'].get('launches_per_step_profiled'))"
cd /tmp && export TMPDIR=/tmp
JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r4_final_llff_noaux -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP > /dev/null 2>&1
python3 $R/tools/prof_summary.py $R/$O/r4_final_llff_noaux/k_kernel_stats.csv 30 27 > $R/$O/r4_final_llff_trace_noaux_summary.txt; rm -rf $R/$O/r4_final_llff_noaux/*kernel_trace.csv
cd $R
timeout 900 python3 tools/converge.py --compress 1 --graph --image-size 400 --views 100 > $O/r4_final_full_schedule.jsonl 2> $O/r4_final_full_schedule.err; tail -1 $O/r4_final_full_schedule.jsonl | cut -c1-400
