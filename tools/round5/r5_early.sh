#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 600 python tools/converge.py --compress 1 --image-size 400 --views 100 --graph --max-iter 3000 --report-every 500 2>&1 | grep '^{"it"' | python -c "
import sys,json
print([ (json.loads(l)['it'], json.loads(l).get('ms_per_iter')) for l in sys.stdin])"; }
run A=1
run JT_BLUR_MFMA=0
run JT_BF16X3=3
run JT_NO_AUX=1
