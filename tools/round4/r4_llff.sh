#!/bin/bash
# LLFF final grid: step time + kernel trace, auxiliary stream on / off
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r4_llff}; shift
cd /tmp && export TMPDIR=/tmp
export JT_TIME_WALK=1
B="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP"
for aux in 0 1; do
  JT_NO_AUX=$aux python3 $B > $O/${TAG}_noaux$aux.json 2> $O/${TAG}_noaux$aux.err
  python3 - <<PY
import json
j=json.loads([l for l in open("$O/${TAG}_noaux$aux.json") if l.startswith("{")][-1])
r=j["roofline"]; d=r.get("density_backward") or {}; f=r.get("forward") or {}
print("LLFF noaux=$aux step %.3f ms | shade_bwd %.3f (%.2f) at %.0f samples | fwd %.3f | density bwd %.3f (%.2f)" % (j["ms_per_step"], r["launch_ms"], r["frac"], r["samples_per_launch"], f.get("launch_ms",0), d.get("launch_ms",0), d.get("frac",0)))
PY
  JT_NO_AUX=$aux rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_noaux$aux -o k -- python3 $B > $O/${TAG}_trace_noaux$aux.log 2>&1
  python3 $R/tools/prof_summary.py $O/${TAG}_trace_noaux$aux/k_kernel_stats.csv 40 27 > $O/${TAG}_trace_noaux${aux}_summary.txt
  head -22 $O/${TAG}_trace_noaux${aux}_summary.txt | cut -c1-140
  rm -rf $O/${TAG}_trace_noaux$aux/*kernel_trace.csv
done
