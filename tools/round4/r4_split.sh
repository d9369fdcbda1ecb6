#!/bin/bash
# round 4: the split appearance backward (JT_BWD_SPLIT = run length) against the fused kernel: parity, step time, kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
export JT_TIME_WALK=1
for s in 8 16 32; do
  JT_BWD_SPLIT=$s timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q > $O/r4_split_parity_$s.log 2>&1
  echo "parity split=$s rc=$? $(tail -1 $O/r4_split_parity_$s.log)"
done
JT_BWD_SPLIT=16 timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp_400cube or llff_final_grid" > $O/r4_split_fullsize_16.log 2>&1
echo "fullsize split=16 rc=$? $(tail -1 $O/r4_split_fullsize_16.log)"
for s in 0 8 16 32; do
  for aux in 0 1; do
    JT_NO_AUX=$aux JT_BWD_SPLIT=$s python3 bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras > $O/r4_split_bench_${s}_noaux$aux.json 2> $O/r4_split_bench_${s}_noaux$aux.err
    python3 - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r4_split_bench_${s}_noaux$aux.json") if l.startswith("{")][-1])
    r=j.get("roofline",{}); d=r.get("density_backward") or {}; f=r.get("forward") or {}
    print("split=%-3s noaux=%s step %.3f ms | shade_bwd %.3f (%.2f) fwd %.3f | density bwd %.3f ms" % ("$s","$aux",j["ms_per_step"],r.get("launch_ms",0),r.get("frac",0),f.get("launch_ms",0),d.get("launch_ms",0)))
except Exception as e:
    print("split=$s noaux=$aux FAILED", e)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for s in 8 16 32; do
  JT_NO_AUX=1 JT_BWD_SPLIT=$s rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_split_trace_$s -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras > $O/r4_split_trace_$s.log 2>&1
  python3 $R/tools/prof_summary.py $O/r4_split_trace_$s/k_kernel_stats.csv 14 27 > $O/r4_split_trace_${s}_summary.txt
  head -14 $O/r4_split_trace_${s}_summary.txt | cut -c1-150
  rm -rf $O/r4_split_trace_$s/*kernel_trace.csv
done
