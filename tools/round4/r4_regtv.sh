#!/bin/bash
# TV regulariser kernels walking rows (vertical neighbours in registers): tests + LLFF kernel times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_guards.py -x -q 2>&1 | tail -2
python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "llff" 2>&1 | tail -2
python3 -m pytest tests/test_gpu_graph.py tests/test_gpu_trajectory.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
L="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP"
for i in 1 2; do python3 $L 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('LLFF step %.3f ms' % j['ms_per_step'])"; done
JT_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_regtv -o k -- python3 $L > /dev/null 2>&1
python3 $R/tools/prof_summary.py $O/r4_regtv/k_kernel_stats.csv 40 27 | grep -E "total|k_reg_batch"
rm -rf $O/r4_regtv/*kernel_trace.csv
