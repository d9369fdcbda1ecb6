#!/bin/bash
# forward-facing scenes for the joint optimisation: which synthetic capture lets bat_llff_VM_MLP's (compressed) schedule recover
# the camera centres >= 5 x?  (the HIP path; its pose-error curve equals the oracle loop's, tests/test_gpu_convergence.py)
cd $GRAFT_REPO_ROOT
BASE="--config bat_llff_VM_MLP --views 40 --image-size 240 --llff-focus 0.0 --gt-z-range 0.35,0.6 --gt-stairs 8 --gt-blobs 6 --gt-blob-radius 0.15,0.35 --graph"
run() { tag=$1; shift; timeout 600 python tools/converge.py $BASE "$@" 2>&1 | grep final | cut -c1-400 | sed "s/^/$tag /" >> gpurun_out/r5_llff_scenes.txt; }
: > gpurun_out/r5_llff_scenes.txt
run base_b0.3 --llff-baseline 0.3
run b0.6 --llff-baseline 0.6
run b1.0 --llff-baseline 1.0
run b0.6_zs0.5 --llff-baseline 0.6 --llff-zspread 0.5
run b0.6_blobs16 --llff-baseline 0.6 --gt-blobs 16
run b0.6_stairs16 --llff-baseline 0.6 --gt-stairs 16
run b0.6_img360 --llff-baseline 0.6 --image-size 360
run b0.6_views80 --llff-baseline 0.6 --views 80
cat gpurun_out/r5_llff_scenes.txt
