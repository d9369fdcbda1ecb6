#!/bin/bash
cd $GRAFT_REPO_ROOT
JT_TILE_CFG=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tile" 2>&1 | tail -n 5 > gpurun_out/r5_tile5_parity1.log
JT_BWD_SPLIT=1 JT_TILE_CFG=1 bash tools/kstat.sh r5t5_cfg1 > gpurun_out/r5_tile5_ks1.txt 2>&1
JT_BWD_SPLIT=1 JT_TILE_CFG=0 bash tools/kstat.sh r5t5_cfg0 > gpurun_out/r5_tile5_ks0.txt 2>&1
tail -n 3 gpurun_out/r5_tile5_parity1.log
for f in gpurun_out/r5_tile5_ks1.txt gpurun_out/r5_tile5_ks0.txt; do echo "== $f"; grep -E "total kernel|k_tile|k_shade_bwd" $f | cut -c1-130; done
