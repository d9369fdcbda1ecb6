#!/bin/bash
# bench.py (secondary legs off) under several builds of the library: tools/bench_variants.sh <tag> "<variant names>" [bench flags]
# variant "default" = the in-tree libjt_render.so; others = joint_tensorf_amd/lib/variants/<name>.so (tools/build_variant.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; VARS=$2; shift; shift
export JT_TIME_WALK=1
for v in $VARS; do
  if [ "$v" = default ]; then unset JT_LIB_PATH; else export JT_LIB_PATH=$R/joint_tensorf_amd/lib/variants/$v.so; fi
  python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_$v.json 2> $R/gpurun_out/${TAG}_$v.err
  python3 - <<PY
import json
try:
    j=json.loads([l for l in open("$R/gpurun_out/${TAG}_$v.json") if l.startswith("{")][-1])
    r=j.get("roofline",{}); d=r.get("density_backward") or {}; f=r.get("forward") or {}
    print("%-10s %-10s step %.3f ms | shade_bwd %.3f (%.2f) fwd %.3f | density bwd %.3f ms (%.2f) listed %.0f" % ("$TAG","$v",j["ms_per_step"],r.get("launch_ms",0),r.get("frac",0),f.get("launch_ms",0),d.get("launch_ms",0),d.get("frac",0),d.get("samples_per_launch",0)))
except Exception as e:
    print("$TAG $v FAILED", e)
PY
done
