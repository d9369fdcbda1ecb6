#!/bin/bash
# finer sweep of the scatter's workgroup cap with the forked GEMMs (the CUs the scatter leaves are the GEMMs')
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(round(j['ms_per_step'],3), 'scatter', round(r.get('launch_ms_scatter',0),3))"; }
for rep in 1 2 3; do
for s in 256 224 208 200 192 184 176; do
  echo "rep $rep JT_SCATTER_WGS=$s: $(JT_SCATTER_WGS=$s $B 2>/dev/null | get)"
done; done
