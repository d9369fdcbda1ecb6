#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_guards.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_matrix_mode.py -x -q > $O/r4_t1_a.log 2>&1; echo "a rc=$? $(tail -1 $O/r4_t1_a.log)"
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp_400cube or llff_final_grid" > $O/r4_t1_b.log 2>&1; echo "b rc=$? $(tail -1 $O/r4_t1_b.log)"
grep -E "FAILED|Error|assert" $O/r4_t1_a.log $O/r4_t1_b.log | head -20
