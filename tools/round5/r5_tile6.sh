#!/bin/bash
cd $GRAFT_REPO_ROOT
JT_TILE_CFG=1 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile" 2>&1 | tail -n 8 > gpurun_out/r5_tile6_full1.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile" 2>&1 | tail -n 8 > gpurun_out/r5_tile6_full0.log
tail -n 4 gpurun_out/r5_tile6_full1.log gpurun_out/r5_tile6_full0.log
