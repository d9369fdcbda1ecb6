#!/bin/bash
# the whole GPU suite as the driver runs it, with timing
cd $GRAFT_REPO_ROOT
( time timeout 3400 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r5_fulltests.log 2>&1
tail -n 40 gpurun_out/r5_fulltests.log
