#!/bin/bash
# the per-stage launch-mode choice of GraphedTrainStep: graph tests, then the full bat_blender_VM schedule and a compressed LLFF
# schedule with the choice on and off (JT_GRAPH_ADAPTIVE=0: replay wherever a graph exists, rounds 2-4's behaviour)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_graph.py -x -q 2>&1 | tail -15
python tools/converge.py --compress 40 --image-size 200 --views 40 --graph > /dev/null 2>&1   # warms the box
for a in 1 0; do
echo "== blender full schedule, adaptive=$a"
JT_GRAPH_ADAPTIVE=$a timeout 1500 python tools/converge.py --compress 1 --image-size 400 --views 100 --graph 2>&1 | grep '^{' > gpurun_out/r5_adaptive_blender_$a.jsonl
tail -n 2 gpurun_out/r5_adaptive_blender_$a.jsonl | cut -c1-900
echo "== llff schedule / 4, adaptive=$a"
JT_GRAPH_ADAPTIVE=$a timeout 1500 python tools/converge.py --config bat_llff_VM_MLP --compress 4 --image-size 0 --views 20 --graph 2>&1 | grep '^{' > gpurun_out/r5_adaptive_llff_$a.jsonl
tail -n 2 gpurun_out/r5_adaptive_llff_$a.jsonl | cut -c1-900
done
