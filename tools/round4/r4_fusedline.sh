#!/bin/bash
# the fused 20-channel backward with plane 0's z line in LDS against the split form: parity + LLFF step and kernel times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py tests/test_gpu_units.py -x -q 2>&1 | tail -2
python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "llff" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
export JT_TIME_WALK=1
L="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP"
for v in "JT_X=0" "JT_BWD_LDS_LINE=-1"; do
  for i in 1 2; do env $v python3 $L 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; print('$v LLFF step %.3f ms  bwd %.3f frac %.3f' % (j['ms_per_step'], r['launch_ms'], r['frac']))"; done
  env $v JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_fl_$v -o k -- python3 $L > $O/r4_fl_$v.log 2>&1
  python3 $R/tools/prof_summary.py $O/r4_fl_$v/k_kernel_stats.csv 40 27 | grep -E "total|k_shade_scatter|k_shade_bwd|k_wgrad"
  rm -rf $O/r4_fl_$v/*kernel_trace.csv
done
