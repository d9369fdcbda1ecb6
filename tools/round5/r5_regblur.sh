#!/bin/bash
# fused regulariser value + gradient, blur on the matrix cores: unit tests, trajectories, timings
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_units.py tests/test_gpu_trajectory.py tests/test_gpu_graph.py tests/test_gpu_lifecycle.py -x -q 2>&1 | tail -n 12 > gpurun_out/r5_regblur_tests.log
JT_FUSE_REG=0 bash tools/kstat.sh r5rb_llff_unfused --config bat_llff_VM_MLP > gpurun_out/r5_regblur_llff_unfused.txt 2>&1
bash tools/kstat.sh r5rb_llff --config bat_llff_VM_MLP > gpurun_out/r5_regblur_llff.txt 2>&1
JT_BLUR_MFMA=0 bash tools/kstat.sh r5rb_blur_old --stage 4 --it 9000 > gpurun_out/r5_regblur_blur_old.txt 2>&1
bash tools/kstat.sh r5rb_blur --stage 4 --it 9000 > gpurun_out/r5_regblur_blur.txt 2>&1
tail -n 6 gpurun_out/r5_regblur_tests.log
for f in llff_unfused llff blur_old blur; do echo "== $f"; grep -E "total kernel|k_reg|k_blur|k_adam" gpurun_out/r5_regblur_$f.txt | cut -c1-130; grep -h '"value"' gpurun_out/ks_r5rb_$f.log | cut -c80-330; done
