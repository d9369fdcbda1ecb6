#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP"
for cfg in "JT_BWD_SPLIT=0" "JT_BWD_SPLIT=16 JT_SCATTER_FLAGS=0" "JT_BWD_SPLIT=16 JT_SCATTER_FLAGS=1" "JT_BWD_SPLIT=8 JT_SCATTER_FLAGS=0" "JT_BWD_SPLIT=8 JT_SCATTER_FLAGS=1"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4l2_$tag -o k -- python3 $B > $O/r4l2_$tag.log 2>&1
  echo "== $cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/r4l2_$tag.log | head -1)"
  python3 $R/tools/prof_summary.py $O/r4l2_$tag/k_kernel_stats.csv 40 27 | grep -E "k_shade_scatter|k_shade_bwd|k_march_bwd_walk" | cut -c1-130
  rm -rf $O/r4l2_$tag/*kernel_trace.csv
done
