#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe --config bat_llff_VM_MLP --it 30000"
for rep in 1 2 3; do
JT_GRAPH=1 timeout 300 python bench.py $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('llff it30000 replayed', round(j['ms_per_step'], 4), j['config'].get('launch'))"
JT_BENCH_SAME_STATE=1 timeout 300 python bench.py $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('llff it30000 eager, same state', round(j['ms_per_step'], 4), j['config'].get('launch'))"
done
