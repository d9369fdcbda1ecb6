#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_trace.py -q -s -m gpu 2>&1 | grep -v Warning | tail -60 > gpurun_out/r6e_engine.txt
timeout 1200 python -m pytest tests/test_gpu_graph.py tests/test_gpu_trajectory.py tests/test_gpu_lifecycle.py -x -q 2>&1 | tail -8 > gpurun_out/r6e_graph.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
$B > gpurun_out/r6e_bench_lean.json 2>/dev/null
$B --config bat_llff_VM_MLP --stage 0 > gpurun_out/r6e_bench_llff0_lean.json 2>/dev/null
cat gpurun_out/r6e_engine.txt gpurun_out/r6e_graph.txt
