#!/bin/bash
# round 6, run aa: the backward on the calling thread + the planned Adam step: tests, then eager steps A/B
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_units.py tests/test_gpu_graph.py tests/test_gpu_trajectory.py tests/test_engine_trace.py tests/test_gpu_dist.py tests/test_gpu_guards.py tests/test_gpu_eval.py tests/test_gpu_lifecycle.py -x -q -m gpu 2>&1 | tail -4
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('$1', round(json.loads(l)['ms_per_step'], 4))"; }
for rep in 1 2; do
for t in 1 0; do
  JT_AUTOGRAD_THREAD=$t JT_ADAM_PLAN=$((1-t)) timeout 300 python bench.py $B --scene fitted --steps 300 --warmup 20 2>/dev/null | line "fitted eager   worker-thread=$t"
  JT_AUTOGRAD_THREAD=$t JT_ADAM_PLAN=$((1-t)) timeout 300 python bench.py $B --scene blobs --steps 300 --warmup 20 2>/dev/null | line "blobs eager    worker-thread=$t"
  JT_AUTOGRAD_THREAD=$t JT_ADAM_PLAN=$((1-t)) timeout 300 python bench.py $B --stage 0 --steps 200 --warmup 20 2>/dev/null | line "stage 0 eager  worker-thread=$t"
  JT_AUTOGRAD_THREAD=$t JT_ADAM_PLAN=$((1-t)) timeout 300 python bench.py $B --steps 40 --warmup 5 2>/dev/null | line "headline       worker-thread=$t"
  JT_AUTOGRAD_THREAD=$t JT_ADAM_PLAN=$((1-t)) timeout 300 python bench.py $B --config bat_llff_VM_MLP --steps 40 --warmup 5 2>/dev/null | line "llff final     worker-thread=$t"
done; done
