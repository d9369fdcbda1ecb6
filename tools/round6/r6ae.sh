#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
for st in "20 5" "300 20"; do set -- $st
JT_GRAPH=1 timeout 300 python bench.py $B --scene fitted --steps $1 --warmup $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('fitted replayed steps $1', round(j['ms_per_step'], 4), j['config'].get('launch'), j['config'].get('graph_stats'))"
done
JT_AUTOGRAD_THREAD=1 JT_GRAPH=1 timeout 300 python bench.py $B --scene fitted --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('worker thread, replayed steps 20', round(j['ms_per_step'], 4))"
