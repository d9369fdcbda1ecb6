#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
B="bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras"
line() { python3 - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-40s step %.3f ms" % (sys.argv[2], j["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for cfg in "JT_BWD_SPLIT=0" "JT_BWD_SPLIT=16" "JT_BWD_SPLIT=16 JT_SCATTER_FIRST=1" "JT_BWD_SPLIT=16 JT_SCATTER_BLOCKS_PER_CU=1" "JT_BWD_SPLIT=16 JT_SCATTER_BLOCKS_PER_CU=3" "JT_BWD_SPLIT=8" "JT_BWD_SPLIT=32"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg python3 $B > $O/r4s3_$tag.json 2> $O/r4s3_$tag.err
  line $O/r4s3_$tag.json "$cfg"
done
cd /tmp && export TMPDIR=/tmp
JT_BWD_SPLIT=16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4s3_trace16 -o k -- python3 $R/$B > $O/r4s3_trace16.log 2>&1
python3 $R/tools/prof_summary.py $O/r4s3_trace16/k_kernel_stats.csv 12 27 | cut -c1-130
rm -rf $O/r4s3_trace16/*kernel_trace.csv
