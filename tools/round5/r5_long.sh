#!/bin/bash
cd $GRAFT_REPO_ROOT
JT_LONG_TESTS=1 timeout 3300 python -m pytest tests/test_gpu_convergence.py -x -q -s -k "full_schedule_with_the_oracle" 2>&1 | grep -E "^hip|^oracle|passed|failed|Error" | cut -c1-900 > gpurun_out/r5_llff_oracle_full.log
cat gpurun_out/r5_llff_oracle_full.log
