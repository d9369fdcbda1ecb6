#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_units.py tests/test_gpu_trajectory.py tests/test_gpu_graph.py tests/test_gpu_lifecycle.py -x -q 2>&1 | tail -n 8 > gpurun_out/r5_blur2_tests.log
bash tools/kstat.sh r5b2_blur --stage 4 --it 9000 > gpurun_out/r5_blur2_blur.txt 2>&1
tail -n 4 gpurun_out/r5_blur2_tests.log
grep -E "total kernel|k_blur" gpurun_out/r5_blur2_blur.txt | cut -c1-130
