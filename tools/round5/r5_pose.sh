#!/bin/bash
# pose-only backward without walkers (k_shade_bwd<SPLIT> + k_pose_gather) against the fused kernel with its scatter muted
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_edge.py tests/test_gpu_eval.py tests/test_gpu_fused.py tests/test_gpu_units.py -x -q 2>&1 | tail -n 6 > gpurun_out/r5_pose_tests.log
JT_POSE_BWD=0 python tools/eval_bench.py --no-render --graph --test-iters 100 --batch-views 32 2>&1 | grep -v amdgpu.ids | tail -n 4 > gpurun_out/r5_pose_old.txt
python tools/eval_bench.py --no-render --graph --test-iters 100 --batch-views 32 2>&1 | grep -v amdgpu.ids | tail -n 4 > gpurun_out/r5_pose_new.txt
python tools/eval_bench.py --no-render --graph --test-iters 100 --batch-views 32 --scene blobs 2>&1 | grep -v amdgpu.ids | tail -n 4 > gpurun_out/r5_pose_new_blobs.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_r5pose -o k -- python3 $GRAFT_REPO_ROOT/tools/eval_bench.py --no-render --graph --test-iters 100 > $GRAFT_REPO_ROOT/gpurun_out/ks_r5pose.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py gpurun_out/ks_r5pose/k_kernel_stats.csv 12 1 > gpurun_out/r5_pose_trace.txt
tail -n 4 gpurun_out/r5_pose_tests.log; cat gpurun_out/r5_pose_old.txt gpurun_out/r5_pose_new.txt gpurun_out/r5_pose_new_blobs.txt | cut -c1-600; cat gpurun_out/r5_pose_trace.txt | cut -c1-150
