#!/bin/bash
# the same forward-facing scene under less compressed schedules (compress 1 = bat_llff_VM_MLP's own 50 000 iterations)
cd $GRAFT_REPO_ROOT
BASE="--config bat_llff_VM_MLP --views 40 --image-size 240 --llff-focus 0.0 --gt-z-range 0.35,0.6 --gt-stairs 8 --gt-blobs 6 --gt-blob-radius 0.15,0.35 --graph --llff-baseline 0.3"
run() { tag=$1; shift; timeout 1500 python tools/converge.py $BASE "$@" 2>&1 | grep '^{' | cut -c1-420 | sed "s/^/$tag /" >> gpurun_out/r5_llff_scenes2.txt; }
: > gpurun_out/r5_llff_scenes2.txt
run c4 --compress 4
run c2 --compress 2
run c1 --compress 1
run c2_seed1 --compress 2 --seed 1
grep final gpurun_out/r5_llff_scenes2.txt
