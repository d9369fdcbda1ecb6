#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 3400 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_convergence.py ) > gpurun_out/r5_resttests.log 2>&1
tail -n 8 gpurun_out/r5_resttests.log
