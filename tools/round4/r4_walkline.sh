#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in bat_blender_VM bat_llff_VM_MLP; do
for v in default noline; do
  if [ $v = default ]; then unset JT_LIB_PATH; else export JT_LIB_PATH=$R/joint_tensorf_amd/lib/variants/$v.so; fi
  JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4wl_${c}_$v -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config $c > $O/r4wl_${c}_$v.log 2>&1
  echo "== $c $v"; python3 $R/tools/prof_summary.py $O/r4wl_${c}_$v/k_kernel_stats.csv 40 27 | grep -E "k_shade_bwd|k_march_bwd_walk" | cut -c1-130
  rm -rf $O/r4wl_${c}_$v/*kernel_trace.csv
done; done
