#!/bin/bash
# Evidence for one bench.py command on the GPU box: (1) the plain JSON line, (2) rocprofv3 kernel trace + stats, (3)/(4) two
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE; gpurun wants --pmc runs to carry --kernel-trace only), then the summaries.
#   tools/profile_cmd.sh <tag> [bench.py flags ...]      -> gpurun_out/<tag>_*
# Copy what is to be judged into profiles/ (round-tagged names).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
CMD="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras $*"
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export JT_TIME_WALK=1
python3 $CMD > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
# (the trace without the in-step timing of the density walk: two launches per iteration that count its listed samples; bench.py
#  still turns it on by itself for every configuration but the default one)
JT_TIME_WALK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o k -- python3 $CMD > $O/${TAG}_trace.log 2>&1
if [ -z "$NO_PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_FETCH_SIZE -o p -- python3 $CMD > $O/${TAG}_pmc_FETCH_SIZE.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_WRITE_SIZE -o p -- python3 $CMD > $O/${TAG}_pmc_WRITE_SIZE.log 2>&1
fi
cd $R
NL=$(python3 -c "import json,sys; j=json.loads([l for l in open('$O/${TAG}_bench_line.json') if l.startswith('{')][-1]); print(j['roofline'].get('process_launches', j['steps']))" 2>/dev/null || echo 1)
python3 tools/prof_summary.py $O/${TAG}_trace/k_kernel_stats.csv 30 $NL > $O/${TAG}_trace_summary.txt
if [ -z "$NO_PMC" ]; then
python3 tools/pmc_summary.py $O/${TAG}_pmc_FETCH_SIZE/p_counter_collection.csv k_shade k_march k_wgrad k_blur k_adam > $O/${TAG}_pmc_FETCH_SIZE_summary.txt
python3 tools/pmc_summary.py $O/${TAG}_pmc_WRITE_SIZE/p_counter_collection.csv k_shade k_march k_wgrad k_blur k_adam > $O/${TAG}_pmc_WRITE_SIZE_summary.txt
python3 tools/pmc_traffic_instep.py $O/${TAG}_pmc_FETCH_SIZE/p_counter_collection.csv $O/${TAG}_pmc_WRITE_SIZE/p_counter_collection.csv $O/${TAG}_bench_line.json $O/${TAG}_pmc_traffic.json > /dev/null
fi
head -16 $O/${TAG}_trace_summary.txt
[ -z "$NO_PMC" ] && cat $O/${TAG}_pmc_traffic.json
# launches of ONE iteration, from the ordered trace: iterations are delimited by k_pose_fwd (the first launch of a training
# iteration); the per-kernel stats above also count the process's set-up (one upload per view of the image set, parameter and
# moment fills) and its priming iterations
python3 - > $O/${TAG}_launches_per_iteration.json <<PY
import csv, glob, json, statistics
rows = []
for f in glob.glob("$O/${TAG}_trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"]))
rows.sort()
cuts = [i for i, (_, n) in enumerate(rows) if "k_pose_fwd" in n]
per = [b - a for a, b in zip(cuts, cuts[1:])]
tail = per[len(per) // 2:] or per or [0]
print(json.dumps(dict(launches_per_iteration_median=statistics.median(tail), min=min(tail), max=max(tail), iterations=len(tail),
                      all_launches_of_the_process=len(rows), delimiter="k_pose_fwd")))
PY
cat $O/${TAG}_launches_per_iteration.json
rm -rf $O/${TAG}_trace/*kernel_trace.csv 2>/dev/null   # keep the merged-back set small: stats + pmc csvs stay
true
