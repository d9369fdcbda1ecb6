#!/bin/bash
# workgroups per tensor in the regulariser forward
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_units.py -x -q -k "reg or tv or l1" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for nb in 512 1024; do
for cfg in bat_blender_VM bat_llff_VM_MLP; do
  JT_REG_BLOCKS=$nb JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_reg_${nb}_$cfg -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config $cfg > $O/r4_reg_${nb}_$cfg.log 2>&1
  echo "== blocks $nb $cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/r4_reg_${nb}_$cfg.log | head -1)"
  python3 $R/tools/prof_summary.py $O/r4_reg_${nb}_$cfg/k_kernel_stats.csv 40 27 | grep -E "k_reg_batch"
  rm -rf $O/r4_reg_${nb}_$cfg/*kernel_trace.csv
done; done
