#!/bin/bash
# second sweep of forward-facing scene variants around the scene of the tests (s3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
run() { tag=$1; shift
  timeout 600 python3 tools/converge.py --config bat_llff_VM_MLP --compress 10 --image-size 240 --graph "$@" > $O/r4_llffconv_$tag.log 2>&1
  echo "== $tag $*"; grep '"final"' $O/r4_llffconv_$tag.log | cut -c1-330 || tail -3 $O/r4_llffconv_$tag.log
}
S3="--gt-z-range 0.35,0.6 --gt-stairs 8 --gt-blobs 6 --gt-blob-radius 0.15,0.35"
run u1 --llff-baseline 0.2 $S3
run u2 --llff-baseline 0.3 $S3 --image-size 320
run u3 --llff-baseline 0.3 --gt-z-range 0.35,0.6 --gt-stairs 12 --gt-blobs 3 --gt-blob-radius 0.15,0.35
run u4 --llff-baseline 0.3 $S3 --seed 1
run u5 --llff-baseline 0.3 $S3 --seed 2
run u6 --llff-baseline 0.3 $S3 --views 60
run u7 --llff-baseline 0.3 --gt-z-range 0.3,0.5 --gt-stairs 8 --gt-blobs 6 --gt-blob-radius 0.2,0.45 --gt-stairs-near 0.5
run u8 --llff-baseline 0.3 $S3 --test-iter 60
