#!/bin/bash
# round 6, run ap: same-box A/B of the scan kernel change (previous build as joint_tensorf_amd/lib/libjt_render_prev.so via JT_LIB_PATH)
cd $GRAFT_REPO_ROOT
B="--no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
P=$GRAFT_REPO_ROOT/joint_tensorf_amd/lib/libjt_render_prev.so
N=$GRAFT_REPO_ROOT/joint_tensorf_amd/lib/libjt_render.so
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('$1', round(json.loads(l)['ms_per_step'], 4))"; }
for rep in 1 2 3; do for L in prev new; do
  if [ $L = prev ]; then LP=$P; else LP=$N; fi
  JT_LIB_PATH=$LP timeout 300 python bench.py $B --config bat_llff_VM_MLP --steps 40 --warmup 5 2>/dev/null | line "$L llff"
  JT_LIB_PATH=$LP JT_BENCH_SAME_STATE=1 timeout 300 python bench.py $B --config bat_llff_VM_MLP --it 30000 2>/dev/null | line "$L llff it30000 eager"
  JT_LIB_PATH=$LP timeout 300 python bench.py $B --steps 40 --warmup 5 2>/dev/null | line "$L headline"
done; done
