#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_units.py tests/test_gpu_guards.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py tests/test_gpu_graph.py tests/test_gpu_trajectory.py -x -q > $O/r4_t2_a.log 2>&1; echo "a rc=$? $(tail -1 $O/r4_t2_a.log)"
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp_400cube or llff_final_grid or llff_stage0" > $O/r4_t2_b.log 2>&1; echo "b rc=$? $(tail -1 $O/r4_t2_b.log)"
grep -E "FAILED|Error|assert" $O/r4_t2_a.log $O/r4_t2_b.log | head -20
export JT_TIME_WALK=1
for c in bat_blender_VM bat_llff_VM_MLP; do
  for env in "X=1" "JT_WALK_LDS_LINE=0"; do
  env $env python3 bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config $c > $O/r4_t2_$c.json 2> $O/r4_t2_$c.err
  python3 - <<PY
import json
j=json.loads([l for l in open("$O/r4_t2_$c.json") if l.startswith("{")][-1])
r=j["roofline"]; d=r.get("density_backward") or {}; f=r.get("forward") or {}
print("$c $env step %.3f ms | shade_bwd %.3f (%.2f) | fwd %.3f | density bwd %.3f (%.2f)" % (j["ms_per_step"], r["launch_ms"], r["frac"], f.get("launch_ms",0), d.get("launch_ms",0), d.get("frac",0)))
PY
  done
done
