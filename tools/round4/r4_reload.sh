#!/bin/bash
# walkers fetch only the taps of slots whose texel changed: parity + kernel times (JT_WALK_RELOAD=1 variant = all six always)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py tests/test_gpu_eval.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
export JT_NO_AUX=1
for v in "" reload; do
for cfg in bat_blender_VM bat_llff_VM_MLP; do
  if [ -n "$v" ]; then [ -f $R/joint_tensorf_amd/lib/variants/$v.so ] || continue; export JT_LIB_PATH=$R/joint_tensorf_amd/lib/variants/$v.so; else unset JT_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_rl_${v:-new}_$cfg -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config $cfg > $O/r4_rl_${v:-new}_$cfg.log 2>&1
  echo "== ${v:-new} $cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/r4_rl_${v:-new}_$cfg.log | head -1)"
  python3 $R/tools/prof_summary.py $O/r4_rl_${v:-new}_$cfg/k_kernel_stats.csv 40 27 | grep -E "k_shade_bwd|k_shade_scatter|k_march_bwd_walk"
  rm -rf $O/r4_rl_${v:-new}_$cfg/*kernel_trace.csv
done; done
unset JT_LIB_PATH JT_NO_AUX
for i in 1 2; do python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dense step %.3f ms  %.0f rays/s  bwd %.3f' % (j['ms_per_step'], j['value'], j['roofline']['launch_ms']))"; done
