#!/bin/bash
# round 6, run ac: pose-only march with stored density derivatives: tests, then test-time optimisation A/B
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_edge.py tests/test_gpu_eval.py tests/test_gpu_lifecycle.py tests/test_gpu_fused.py tests/test_abi.py -x -q 2>&1 | tail -4
for rep in 1 2; do for v in 0 1; do
JT_POSE_MARCH=$v timeout 600 python tools/eval_bench.py --graph --no-render 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('JT_POSE_MARCH=$v replayed', d.get('test_time_optim'))"
done; done
JT_POSE_MARCH=1 timeout 600 python tools/eval_bench.py --no-render --test-iters 40 --batch-views 1,8,32 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('batched', d.get('test_time_optim_batched_ms_per_view_iteration'))"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6ac_trace -o k -- python3 $GRAFT_REPO_ROOT/tools/eval_bench.py --no-render --graph --test-iters 100 > $GRAFT_REPO_ROOT/gpurun_out/r6ac.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py gpurun_out/r6ac_trace/k_kernel_stats.csv 12 200
rm -rf gpurun_out/r6ac_trace
