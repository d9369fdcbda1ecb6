#!/bin/bash
# MFMA tile scatter: parity of the small fixtures (both workgroup shapes), steady-state kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tile" 2>&1 | tail -n 15 > gpurun_out/r5_tile3_parity.log
JT_TILE_CFG=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tile" 2>&1 | tail -n 15 > gpurun_out/r5_tile3_parity1.log
JT_BWD_SPLIT=1 bash tools/kstat.sh r5t3_cfg0 > gpurun_out/r5_tile3_ks0.txt 2>&1
JT_BWD_SPLIT=1 JT_TILE_CFG=1 bash tools/kstat.sh r5t3_cfg1 > gpurun_out/r5_tile3_ks1.txt 2>&1
tail -n 6 gpurun_out/r5_tile3_parity.log gpurun_out/r5_tile3_parity1.log
for f in gpurun_out/r5_tile3_ks0.txt gpurun_out/r5_tile3_ks1.txt; do echo "== $f"; grep -E "total kernel|k_tile|k_shade_bwd" $f | cut -c1-130; done
grep -h '"value"' gpurun_out/ks_r5t3_cfg0.log gpurun_out/ks_r5t3_cfg1.log | cut -c1-250
