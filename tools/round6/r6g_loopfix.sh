#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "loop_it" 2>&1 | grep -v Warning | grep 'parity\]\|passed\|failed\|Error\|assert\|Mismatch\|Max ' | head -80 > gpurun_out/r6g_loopfix.txt
cat gpurun_out/r6g_loopfix.txt
