#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_trace.py -q -s -m gpu -k llff 2>&1 | grep 'engine trace' > gpurun_out/r6f_engine.txt
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q 2>&1 | tail -4 >> gpurun_out/r6f_engine.txt
cat gpurun_out/r6f_engine.txt
