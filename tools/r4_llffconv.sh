#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
run() { tag=$1; shift
  timeout 600 python3 tools/converge.py --config bat_llff_VM_MLP --compress 10 --image-size 240 --graph "$@" > $O/r4_llffconv_$tag.log 2>&1
  echo "== $tag $*"; grep '"final"' $O/r4_llffconv_$tag.log | cut -c1-400 || tail -3 $O/r4_llffconv_$tag.log
}
run v1
run v2 --llff-baseline 0.5 --gt-z-range 0.4,0.8 --gt-wall 0.9
run v3 --llff-baseline 0.5 --gt-z-range 0.4,0.8 --gt-wall 0.9 --views 30
run v4 --llff-baseline 0.5 --gt-z-range 0.4,0.8 --gt-wall 0.9 --llff-focus 2.5
run v5 --llff-baseline 0.8 --gt-z-range 0.4,0.8 --gt-wall 0.9
run v6 --llff-baseline 0.5 --gt-z-range 0.4,0.8 --gt-wall 0.9 --compress 5
