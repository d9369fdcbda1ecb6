#!/bin/bash
# fewer scatter / walk workgroups: does the atomic-bound scatter slow down, and do the forked GEMMs use the freed CUs?
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(round(j['ms_per_step'],3), 'chain', round(r.get('launch_ms_chain',0),3), 'scatter', round(r.get('launch_ms_scatter',0),3))"; }
for s in 256 224 192 160 128 96; do
  echo "JT_SCATTER_WGS=$s  noaux: $(JT_NO_AUX=1 JT_SCATTER_WGS=$s $B 2>/dev/null | get)   aux: $(JT_SCATTER_WGS=$s $B 2>/dev/null | get)"
done
for w in 192 144 96; do
  echo "JT_SCATTER_WGS=192 JT_WALK_WGS=$w aux: $(JT_SCATTER_WGS=192 JT_WALK_WGS=$w $B 2>/dev/null | get)"
  echo "JT_SCATTER_WGS=160 JT_WALK_WGS=$w aux: $(JT_SCATTER_WGS=160 JT_WALK_WGS=$w $B 2>/dev/null | get)"
done
