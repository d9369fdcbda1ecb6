#!/bin/bash
# what an iteration of the FINAL stage of a real (converging) run costs: kernel trace of tools/converge.py, last part of it
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/r4_trained -o k -- python3 $R/tools/converge.py --compress 10 --image-size 400 --views 100 --graph > $O/r4_trained.log 2>&1
cd $R
grep '^{' $O/r4_trained.log | cut -c1-300
# 4 000 iterations, the 400^3 grid from iteration 1 100 on: the last 30 % of the trace is ~ iterations 2 800 - 4 000 + evaluation
python3 tools/trace_tail.py $O/r4_trained/k_kernel_trace.csv 0.25 0 40 > $O/r4_trained_tail.txt
head -45 $O/r4_trained_tail.txt
rm -f $O/r4_trained/k_kernel_trace.csv
S3="--gt-z-range 0.35,0.6 --gt-stairs 8 --gt-blobs 6 --gt-blob-radius 0.15,0.35"
python3 tools/converge.py --config bat_llff_VM_MLP --compress 10 --image-size 240 --graph --llff-baseline 0.3 $S3 2>&1 | grep final | cut -c1-400
