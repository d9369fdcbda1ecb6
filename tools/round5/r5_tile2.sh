#!/bin/bash
# tile-owned scatter: parity of the small fixtures, steady-state kernel stats, ablations (JT_TILE_ABL variants)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tile" 2>&1 | tail -15 > gpurun_out/r5_tile2_parity.log
JT_BWD_SPLIT=1 bash tools/kstat.sh r5t2_cfg0 > gpurun_out/r5_tile2_ks0.txt 2>&1
JT_BWD_SPLIT=1 JT_TILE_CFG=1 bash tools/kstat.sh r5t2_cfg1 > gpurun_out/r5_tile2_ks1.txt 2>&1
for v in 1 2 4 7 8; do
  JT_LIB_PATH=$GRAFT_REPO_ROOT/joint_tensorf_amd/lib/variants/tile_abl$v.so JT_BWD_SPLIT=1 bash tools/kstat.sh r5t2_abl$v > gpurun_out/r5_tile2_abl$v.txt 2>&1
done
tail -n 5 gpurun_out/r5_tile2_parity.log
for f in gpurun_out/r5_tile2_ks0.txt gpurun_out/r5_tile2_ks1.txt gpurun_out/r5_tile2_abl*.txt; do echo "== $f"; grep -E "total kernel|k_tile|k_shade_bwd" $f | cut -c1-130; done
grep -h '"value"' gpurun_out/ks_r5t2_cfg0.log gpurun_out/ks_r5t2_cfg1.log | cut -c1-250
