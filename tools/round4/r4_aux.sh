#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
for rep in 1 2; do
for c in bat_blender_VM bat_llff_VM_MLP; do
  for env in "JT_NO_AUX=0" "JT_NO_AUX=1" $EXTRA_ENVS; do
  env $env python3 bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config $c > $O/r4_aux.json 2> $O/r4_aux.err
  python3 - <<PY
import json
j=json.loads([l for l in open("$O/r4_aux.json") if l.startswith("{")][-1])
r=j["roofline"]; d=r.get("density_backward") or {}; f=r.get("forward") or {}
print("$c $env step %.3f ms | shade_bwd %.3f (%.2f) | fwd %.3f | density bwd %.3f" % (j["ms_per_step"], r["launch_ms"], r["frac"], f.get("launch_ms",0), d.get("launch_ms",0)))
PY
  done
done; done
