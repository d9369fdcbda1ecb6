#!/bin/bash
# round 6, run al: the twelve-wave scatter on 224 workgroups (the count that won with eight waves) against its default 192
cd $GRAFT_REPO_ROOT
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
for rep in 1 2 3; do for v in 192 224; do
JT_SCATTER_WGS=$v timeout 300 python bench.py $B --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('JT_SCATTER_WGS=$v headline', round(json.loads(l)['ms_per_step'], 4))"
done; done
