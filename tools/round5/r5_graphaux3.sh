#!/bin/bash
cd $GRAFT_REPO_ROOT
export JT_BENCH_SAME_STATE=1
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],3))"; }
for rep in 1 2 3; do
echo "eager $($B 2>/dev/null | get)  graph $(JT_GRAPH=1 $B 2>/dev/null | get)  graph+fork $(JT_GRAPH=1 JT_GRAPH_AUX=1 $B 2>/dev/null | get)"
done
echo "llff: eager $($B --config bat_llff_VM_MLP --it 30000 2>/dev/null | get)  graph $(JT_GRAPH=1 $B --config bat_llff_VM_MLP --it 30000 2>/dev/null | get)  graph+fork $(JT_GRAPH=1 JT_GRAPH_AUX=1 $B --config bat_llff_VM_MLP --it 30000 2>/dev/null | get)"
