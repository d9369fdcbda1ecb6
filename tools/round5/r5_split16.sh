#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 16 8; do JT_BWD_SPLIT=$v bash tools/kstat.sh r5s_split$v > gpurun_out/r5_split_$v.txt 2>&1; done
bash tools/kstat.sh r5s_fused > gpurun_out/r5_split_fused.txt 2>&1
for v in 16 8; do JT_BWD_SPLIT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | cut -c80-200; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | cut -c80-200
JT_BWD_SPLIT=16 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | cut -c80-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras 2>/dev/null | cut -c80-200
for f in 16 8 fused; do echo "== $f"; grep -E "total kernel|k_shade_bwd|k_shade_scatter" gpurun_out/r5_split_$f.txt | cut -c1-130; done
