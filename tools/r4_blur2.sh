#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_graph.py tests/test_gpu_trajectory.py -x -q -k "blur or parity or fuzz or graph or trajectory or golden" > $O/r4_blur2_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r4_blur2_tests.log)"
grep -E "FAILED|Error" $O/r4_blur2_tests.log | head
VARS=default bash tools/r4_blur.sh
JT_BLUR_LDS=0 VARS=default bash tools/r4_blur.sh 2>&1 | grep -E "^==" 
