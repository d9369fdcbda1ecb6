#!/bin/bash
# first run of the tile-owned scatter: parity of the sixth variant, then kernel stats of the headline with it on
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tile" 2>&1 | tail -15 > gpurun_out/r5_tile1_parity.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile" 2>&1 | tail -15 > gpurun_out/r5_tile1_full.log
JT_BWD_SPLIT=1 bash tools/kstat.sh r5_tile_cfg0 > gpurun_out/r5_tile1_ks0.txt 2>&1
JT_BWD_SPLIT=1 JT_TILE_CFG=1 bash tools/kstat.sh r5_tile_cfg1 > gpurun_out/r5_tile1_ks1.txt 2>&1
bash tools/kstat.sh r5_base > gpurun_out/r5_tile1_ksbase.txt 2>&1
tail -5 gpurun_out/r5_tile1_parity.log gpurun_out/r5_tile1_full.log
head -20 gpurun_out/r5_tile1_ks0.txt gpurun_out/r5_tile1_ks1.txt gpurun_out/r5_tile1_ksbase.txt
grep -h '"value"' gpurun_out/ks_r5_tile_cfg0.log gpurun_out/ks_r5_tile_cfg1.log gpurun_out/ks_r5_base.log | cut -c1-300
