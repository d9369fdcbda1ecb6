#!/bin/bash
# round 6, run as: replayed steps with the weight-gradient GEMMs forked inside the captured graph (JT_GRAPH_AUX=1) against one stream
cd $GRAFT_REPO_ROOT
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('$1', round(json.loads(l)['ms_per_step'], 4))"; }
for rep in 1 2; do for v in 0 1; do
JT_GRAPH=1 JT_GRAPH_AUX=$v timeout 300 python bench.py $B --config bat_llff_VM_MLP --it 30000 2>/dev/null | line "JT_GRAPH_AUX=$v llff it30000 replayed"
JT_GRAPH=1 JT_GRAPH_AUX=$v timeout 300 python bench.py $B 2>/dev/null | line "JT_GRAPH_AUX=$v headline replayed"
done; done
