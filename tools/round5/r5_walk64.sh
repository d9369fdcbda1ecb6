#!/bin/bash
# density walk with the LDS line as doubles (ds_add_f64) against the float copy
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_edge.py tests/test_gpu_parity.py -x -q -k "not tile and not split and not fp32" 2>&1 | tail -n 5 > gpurun_out/r5_walk64_tests.log
JT_WALK_LDS_LINE=1 bash tools/kstat.sh r5w_f32 > gpurun_out/r5_walk64_f32.txt 2>&1
bash tools/kstat.sh r5w_auto > gpurun_out/r5_walk64_auto.txt 2>&1
JT_WALK_LDS_LINE=1 bash tools/kstat.sh r5w_llff_f32 --config bat_llff_VM_MLP > gpurun_out/r5_walk64_llff_f32.txt 2>&1
JT_WALK_LDS_LINE=2 JT_WALK_WAVES=8 bash tools/kstat.sh r5w_llff_f64w8 --config bat_llff_VM_MLP > gpurun_out/r5_walk64_llff_f64w8.txt 2>&1
tail -n 3 gpurun_out/r5_walk64_tests.log
for f in gpurun_out/r5_walk64_f32.txt gpurun_out/r5_walk64_auto.txt gpurun_out/r5_walk64_llff_f32.txt gpurun_out/r5_walk64_llff_f64w8.txt; do echo "== $f"; grep -E "total kernel|k_march_bwd_walk|k_shade_scatter" $f | cut -c1-130; done
