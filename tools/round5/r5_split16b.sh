#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_guards.py tests/test_gpu_graph.py tests/test_gpu_trajectory.py -x -q 2>&1 | tail -n 6 > gpurun_out/r5_split16b_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | cut -c80-200
JT_NO_AUX=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | cut -c80-200
JT_GRAPH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | cut -c80-200
done
tail -n 4 gpurun_out/r5_split16b_tests.log
