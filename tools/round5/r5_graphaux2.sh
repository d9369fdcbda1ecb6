#!/bin/bash
# eager vs replayed vs replayed with the captured fork, over the SAME iterations (JT_BENCH_SAME_STATE=1)
cd $GRAFT_REPO_ROOT
export JT_BENCH_SAME_STATE=1
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
for cfg in "" "--config bat_llff_VM_MLP" "--stage 2" "--config bat_llff_VM_MLP --it 30000"; do
for i in 1 2; do
echo "eager $cfg"; $B $cfg 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['config']['rays_per_iter_per_gpu'], j['config'].get('shaded_samples_per_iter'))"
echo "graph $cfg"; JT_GRAPH=1 $B $cfg 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['config']['rays_per_iter_per_gpu'], j['config'].get('shaded_samples_per_iter'))"
echo "graph_aux $cfg"; JT_GRAPH=1 JT_GRAPH_AUX=1 $B $cfg 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['config']['rays_per_iter_per_gpu'], j['config'].get('shaded_samples_per_iter'))"
done; done
