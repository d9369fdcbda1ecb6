#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/kstat.sh r5rg_parent --n-voxel-final 27000000 --n-rays 4096 > gpurun_out/r5_regress_parent.txt 2>&1
JT_WALK_LDS_LINE=1 bash tools/kstat.sh r5rg_parent_f32 --n-voxel-final 27000000 --n-rays 4096 > gpurun_out/r5_regress_parent_f32.txt 2>&1
JT_FUSE_REG=0 JT_WALK_LDS_LINE=1 bash tools/kstat.sh r5rg_parent_f32_nofuse --n-voxel-final 27000000 --n-rays 4096 > gpurun_out/r5_regress_parent_f32_nofuse.txt 2>&1
for f in parent parent_f32 parent_f32_nofuse; do echo "== $f"; head -14 gpurun_out/r5_regress_$f.txt | cut -c1-120; grep -h '"value"' gpurun_out/ks_r5rg_$f.log | cut -c80-250; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --stage 0 2>/dev/null | cut -c80-260
JT_FUSE_REG=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --stage 0 2>/dev/null | cut -c80-260
