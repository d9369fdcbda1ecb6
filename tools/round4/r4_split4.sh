#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
B="bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras"
line() { python3 - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-50s step %.3f ms" % (sys.argv[2], j["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
for f in 0 1 2 3; do
  JT_SCATTER_FLAGS=$f JT_BWD_SPLIT=16 timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q > $O/r4s4_parity_f$f.log 2>&1
  echo "parity split=16 flags=$f rc=$? $(tail -1 $O/r4s4_parity_f$f.log)"
done
JT_SCATTER_FLAGS=3 JT_BWD_SPLIT=8 timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q > $O/r4s4_parity_8f3.log 2>&1; echo "parity split=8 flags=3 rc=$? $(tail -1 $O/r4s4_parity_8f3.log)"
JT_SCATTER_FLAGS=3 JT_BWD_SPLIT=16 timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp_400cube or llff_final_grid" > $O/r4s4_fullsize.log 2>&1
echo "fullsize split=16 flags=3 rc=$? $(tail -1 $O/r4s4_fullsize.log)"
for cfg in "JT_BWD_SPLIT=0" "JT_BWD_SPLIT=16 JT_SCATTER_FLAGS=0" "JT_BWD_SPLIT=16 JT_SCATTER_FLAGS=1" "JT_BWD_SPLIT=16 JT_SCATTER_FLAGS=2" "JT_BWD_SPLIT=16 JT_SCATTER_FLAGS=3" "JT_BWD_SPLIT=8 JT_SCATTER_FLAGS=3"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg python3 $B > $O/r4s4_$tag.json 2> $O/r4s4_$tag.err
  line $O/r4s4_$tag.json "$cfg"
  env $cfg JT_NO_AUX=1 python3 $B > $O/r4s4_${tag}_noaux.json 2> $O/r4s4_${tag}_noaux.err
  line $O/r4s4_${tag}_noaux.json "$cfg NOAUX"
done
cd /tmp && export TMPDIR=/tmp
for f in 1 3; do
JT_NO_AUX=1 JT_SCATTER_FLAGS=$f JT_BWD_SPLIT=16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4s4_trace_f$f -o k -- python3 $R/$B > $O/r4s4_trace_f$f.log 2>&1
python3 $R/tools/prof_summary.py $O/r4s4_trace_f$f/k_kernel_stats.csv 12 27 | cut -c1-130
rm -rf $O/r4s4_trace_f$f/*kernel_trace.csv
done
