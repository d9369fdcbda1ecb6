#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in bat_blender_VM bat_llff_VM_MLP; do
for env in "JT_NO_AUX=1" "JT_NO_AUX=0" "JT_NO_AUX=1 JT_WALK_LDS_LINE=0"; do
  tag=$(echo ${c}_$env | tr ' =' '__')
  env $env rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4w2_$tag -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config $c > $O/r4w2_$tag.log 2>&1
  echo "== $c $env: $(grep -o '"ms_per_step": [0-9.]*' $O/r4w2_$tag.log | head -1)"; python3 $R/tools/prof_summary.py $O/r4w2_$tag/k_kernel_stats.csv 40 27 | grep -E "k_shade_bwd|k_shade_scatter|k_march_bwd|k_wgrad_b16" | cut -c1-130
  rm -rf $O/r4w2_$tag/*kernel_trace.csv
done; done
