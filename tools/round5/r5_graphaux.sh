#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras"
for i in 1 2; do
echo eager; $B 2>/dev/null | cut -c80-200
echo graph; JT_GRAPH=1 $B 2>/dev/null | cut -c80-200
echo graph_aux; JT_GRAPH=1 JT_GRAPH_AUX=1 $B 2>/dev/null | cut -c80-200
done
echo llff_graph; JT_GRAPH=1 $B --config bat_llff_VM_MLP 2>/dev/null | cut -c80-200
echo llff_graph_aux; JT_GRAPH=1 JT_GRAPH_AUX=1 $B --config bat_llff_VM_MLP 2>/dev/null | cut -c80-200
echo llff_eager; $B --config bat_llff_VM_MLP 2>/dev/null | cut -c80-200
