#!/bin/bash
# 256 against 192 scatter workgroups (24 per XCD: eight CUs per XCD left to the forked GEMMs), alternating, six repeats
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(round(j['ms_per_step'],3), 'scatter', round(r.get('launch_ms_scatter',0),3))"; }
for rep in 1 2 3 4 5 6; do
for s in 256 192; do
  echo "rep $rep JT_SCATTER_WGS=$s: $(JT_SCATTER_WGS=$s $B 2>/dev/null | get)"
done; done
for s in 256 192; do echo "parent yaml JT_SCATTER_WGS=$s: $(JT_SCATTER_WGS=$s $B --n-voxel-final 27000000 --n-rays 4096 2>/dev/null | get)"; done
