#!/bin/bash
# the appearance factors' optimizer step on the auxiliary stream beside the density backward (JT_ADAM_EARLY=0: one launch behind it)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_trajectory.py tests/test_gpu_units.py -x -q 2>&1 | tail -n 3
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],3))"; }
for rep in 1 2 3 4; do
echo "blender        early: $($B 2>/dev/null | get)   off: $(JT_ADAM_EARLY=0 $B 2>/dev/null | get)"
done
for rep in 1 2 3; do
echo "llff it30000   early: $($B --config bat_llff_VM_MLP --it 30000 2>/dev/null | get)   off: $(JT_ADAM_EARLY=0 $B --config bat_llff_VM_MLP --it 30000 2>/dev/null | get)"
done
bash tools/round5/r5_timeline.sh > /dev/null 2>&1
tail -16 gpurun_out/r5_timeline.txt
