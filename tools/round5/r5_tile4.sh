#!/bin/bash
# MFMA tile scatter (one class, eight waves): ablations
cd $GRAFT_REPO_ROOT
for v in 1 2 4 8 15; do
  JT_LIB_PATH=$GRAFT_REPO_ROOT/joint_tensorf_amd/lib/variants/tile_abl$v.so JT_BWD_SPLIT=1 JT_TILE_CFG=1 bash tools/kstat.sh r5t4_abl$v > gpurun_out/r5_tile4_abl$v.txt 2>&1
done
for f in gpurun_out/r5_tile4_abl*.txt; do echo "== $f"; grep -E "k_tile_scatter" $f | cut -c1-130; done
