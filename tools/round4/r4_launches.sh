#!/bin/bash
# fewer launches per iteration (intrinsics copy, Adam coefficient poke, pose Adam, loss head): tests, launch list, step time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_graph.py tests/test_gpu_trajectory.py tests/test_gpu_lifecycle.py tests/test_gpu_guards.py tests/test_gpu_units.py tests/test_gpu_eval.py tests/test_gpu_dist.py tests/test_abi.py -x -q 2>&1 | tail -4
python3 -m pytest tests/test_gpu_convergence.py -x -q 2>&1 | tail -3
bash tools/round4/r4_order.sh > $O/r4_order_list.txt 2>&1; grep -c "us  gap" $O/r4_order_list.txt; grep "us  gap" $O/r4_order_list.txt | awk '{print $6}' | cut -c1-40 | head -40 | tr '\n' ' '
echo
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dense step %.3f ms  %.0f rays/s  bwd %.3f' % (j['ms_per_step'], j['value'], j['roofline']['launch_ms']))"; done
python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('LLFF step %.3f ms' % j['ms_per_step'])"
