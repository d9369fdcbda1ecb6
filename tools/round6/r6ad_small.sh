#!/bin/bash
# round 6, run ad: lattice kernel, frozen-scene regulariser cache, seeded capture: tests + test-time / replayed training timings
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_units.py tests/test_gpu_graph.py tests/test_gpu_eval.py tests/test_gpu_lifecycle.py tests/test_gpu_fused.py tests/test_gpu_edge.py tests/test_gpu_trajectory.py tests/test_abi.py -x -q 2>&1 | tail -4
for rep in 1 2; do
timeout 600 python tools/eval_bench.py --graph --no-render 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('replayed', d.get('test_time_optim'))"
done
timeout 600 python tools/eval_bench.py --no-render 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('eager', d.get('test_time_optim'))"
timeout 600 python tools/eval_bench.py --no-render --test-iters 40 --batch-views 1,8,32 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('batched', d.get('test_time_optim_batched_ms_per_view_iteration'))"
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
JT_GRAPH=1 timeout 300 python bench.py $B --scene fitted --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('fitted replayed', round(json.loads(l)['ms_per_step'], 4))"
