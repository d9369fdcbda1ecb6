#!/bin/bash
# the whole compressed LLFF schedule through the HIP renderer and through the oracle's stock-torch renderer (verdict item 3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
JT_LONG_TESTS=1 timeout 2400 python3 -m pytest tests/test_gpu_convergence.py -x -q -s -k "oracle_render" > $O/r4_llff_oracle_full.log 2>&1
grep -E "^(hip|oracle) |passed|failed|Error|error" $O/r4_llff_oracle_full.log | cut -c1-900 | tail -12
