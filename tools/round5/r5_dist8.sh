#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_dist.py -x -q -k "gpus8 or gpus2" 2>&1 | tail -n 25 > gpurun_out/r5_dist8.log
tail -n 25 gpurun_out/r5_dist8.log
