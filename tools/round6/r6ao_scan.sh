#!/bin/bash
# round 6, run ao: the density backward's scan kernel with two LDS arrays per ray instead of three: parity, then LLFF / headline timing
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_edge.py tests/test_gpu_trajectory.py tests/test_gpu_eval.py -x -q 2>&1 | tail -3
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
for rep in 1 2 3; do
timeout 300 python bench.py $B --config bat_llff_VM_MLP --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('llff', round(json.loads(l)['ms_per_step'], 4))"
JT_BENCH_SAME_STATE=1 timeout 300 python bench.py $B --config bat_llff_VM_MLP --it 30000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('llff it30000 eager', round(json.loads(l)['ms_per_step'], 4))"
timeout 300 python bench.py $B --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('headline', round(json.loads(l)['ms_per_step'], 4))"
done
NO_PMC=1 bash tools/profile_cmd.sh r6ao_llff --config bat_llff_VM_MLP > gpurun_out/r6ao_profile.log 2>&1
grep -i 'scan\|walk\|march_fwd' gpurun_out/r6ao_llff_trace_summary.txt
