#!/bin/bash
# end-of-round evidence on the final code, without the counter passes: GPU suite + smoke, full bench line, LLFF trace, whole schedule
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=gpurun_out
bash tools/round4/r4_fulltests.sh
python3 bench.py > $O/r4_final_bench_full.json 2> $O/r4_final_bench_full.err; python3 -c "
import json; j=json.loads([l for l in open('$O/r4_final_bench_full.json') if l.startswith('{')][-1]); print('bench', j['value'], j['ms_per_step'], j['vs_baseline'], j['roofline']['frac'], j['roofline'].get('traffic_source'), j['roofline'].get('launches_per_step_profiled')); print({k: v.get('ms_per_step', v.get('ms_per_image')) for k, v in j['extra'].items() if isinstance(v, dict)})"
cd /tmp && export TMPDIR=/tmp
bash $R/tools/profile_cmd.sh r4f_default > /dev/null 2>&1; NO_PMC=1 bash $R/tools/profile_cmd.sh r4f_llff --config bat_llff_VM_MLP > /dev/null 2>&1
head -14 $R/$O/r4f_default_trace_summary.txt | cut -c1-130; head -14 $R/$O/r4f_llff_trace_summary.txt | cut -c1-130
cd $R
timeout 900 python3 tools/converge.py --compress 1 --graph --image-size 400 --views 100 > $O/r4_final_full_schedule.jsonl 2> $O/r4_final_full_schedule.err; tail -1 $O/r4_final_full_schedule.jsonl | cut -c1-400
