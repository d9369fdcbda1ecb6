#!/bin/bash
# the round's evidence set once more on the FINAL code (29 launches per iteration): profile set, then the full bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=gpurun_out
bash tools/round4_profiles.sh > $O/r4_final_profiles.log 2>&1; tail -36 $O/r4_final_profiles.log | cut -c1-140
cat $O/r4_default_pipe_busy.json | head -c 900; echo
python3 bench.py > $O/r4_final_bench_full.json 2> $O/r4_final_bench_full.err; python3 -c "
import json; j=json.loads([l for l in open('$O/r4_final_bench_full.json') if l.startswith('{')][-1]); print('bench', j['value'], j['ms_per_step'], j['vs_baseline'], j['roofline']['frac'], j['roofline'].get('traffic_source'), j['roofline'].get('launches_per_step_profiled'))"
