#!/bin/bash
# regulariser / render-loss index arithmetic, LLFF scatter waves: tests + step times + kernel times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_eval.py tests/test_gpu_guards.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
export JT_TIME_WALK=1
L="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP"
run() { tag=$1; shift
  for i in 1 2; do env "$@" python3 $L 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$tag LLFF step %.3f ms' % j['ms_per_step'])"; done
}
run default JT_X=0
run split8_w16 JT_BWD_SPLIT=8
run split8_w8 JT_BWD_SPLIT=8 JT_SCATTER_WAVES=8
export JT_BWD_SPLIT=8
JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_small_llff -o k -- python3 $L > $O/r4_small_llff.log 2>&1
unset JT_BWD_SPLIT
python3 $R/tools/prof_summary.py $O/r4_small_llff/k_kernel_stats.csv 40 27 | grep -E "total|k_reg_batch|k_render_loss|k_shade_scatter|k_shade_bwd"
unset JT_TIME_WALK
B="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras"
for i in 1 2; do python3 $B 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dense step %.3f ms  %.0f rays/s  bwd %.3f' % (j['ms_per_step'], j['value'], j['roofline']['launch_ms']))"; done
JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_small_trace -o k -- python3 $B > $O/r4_small_trace.log 2>&1
python3 $R/tools/prof_summary.py $O/r4_small_trace/k_kernel_stats.csv 40 27 | grep -E "total|k_reg_batch|k_render_loss|k_adam"
rm -rf $O/r4_small_trace/*kernel_trace.csv $O/r4_small_llff/*kernel_trace.csv
