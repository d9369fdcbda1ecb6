#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(round(j['ms_per_step'],3), 'scatter', round(r.get('launch_ms_scatter',0),3))"; }
for rep in 1 2 3; do
for s in 256 192; do
  echo "llff rep $rep JT_SCATTER_WGS=$s: $(JT_SCATTER_WGS=$s $B --config bat_llff_VM_MLP --it 30000 2>/dev/null | get)"
done; done
for rep in 1 2; do
for s in 188 192 196; do
  echo "rep $rep JT_SCATTER_WGS=$s: $(JT_SCATTER_WGS=$s $B 2>/dev/null | get)"
done; done
for s in 256 192; do echo "stage 4 blurred JT_SCATTER_WGS=$s: $(JT_SCATTER_WGS=$s $B --stage 4 --it 9000 2>/dev/null | get)"; done
for s in 256 192; do echo "fitted JT_SCATTER_WGS=$s: $(JT_SCATTER_WGS=$s $B --scene fitted 2>/dev/null | get)"; done
