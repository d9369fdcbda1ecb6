#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
run() { tag=$1; shift
  timeout 600 python3 tools/converge.py --config bat_llff_VM_MLP --compress 10 --image-size 240 --graph "$@" > $O/r4_llffconv_$tag.log 2>&1
  echo "== $tag $*"; grep '"final"' $O/r4_llffconv_$tag.log | cut -c1-330 || tail -3 $O/r4_llffconv_$tag.log
}
S3="--llff-baseline 0.3 --gt-z-range 0.35,0.6 --gt-stairs 8 --gt-blobs 6 --gt-blob-radius 0.15,0.35"
run t1 $S3 --compress 5
run t2 $S3 --llff-focus 3.0
run t3 $S3 --views 30
run t4 $S3 --llff-zspread 0.5
run t5 $S3 --llff-zspread 0.5 --llff-focus 3.0 --compress 5
run t6 --llff-baseline 0.3 --gt-z-range 0.35,0.6 --gt-stairs 10 --gt-blobs 4 --gt-blob-radius 0.2,0.4 --llff-zspread 0.5
