#!/bin/bash
# blurred stages: step time and kernel trace for library variants
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
VARS=${VARS:-"default blurp16 blurp32"}
for v in $VARS; do
  if [ $v = default ]; then unset JT_LIB_PATH; else export JT_LIB_PATH=$R/joint_tensorf_amd/lib/variants/$v.so; fi
  for w in "--stage 4 --it 9000" "--stage 2" "--stage 4"; do
    tag=$(echo ${v}_$w | tr ' -' '__')
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4blur_$tag -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras $w > $O/r4blur_$tag.log 2>&1
    echo "== $v $w: $(grep -o '"ms_per_step": [0-9.]*' $O/r4blur_$tag.log | head -1)"
    python3 $R/tools/prof_summary.py $O/r4blur_$tag/k_kernel_stats.csv 40 27 > $O/r4blur_${tag}_summary.txt
    grep -E "k_blur|total kernel" $O/r4blur_${tag}_summary.txt | cut -c1-130
    rm -rf $O/r4blur_$tag/*kernel_trace.csv
  done
done
