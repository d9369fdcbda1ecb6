#!/bin/bash
cd $GRAFT_REPO_ROOT
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
$B > gpurun_out/r6r_default.json 2>/dev/null
JT_BWD_SPLIT=16 $B > gpurun_out/r6r_split16_explicit.json 2>/dev/null
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 JT_SCATTER_WGS=256 $B > gpurun_out/r6r_w12_wg256.json 2>/dev/null
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 JT_SCATTER_WGS=192 $B > gpurun_out/r6r_w12_wg192.json 2>/dev/null
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 JT_SCATTER_WGS=208 $B > gpurun_out/r6r_w12_wg208.json 2>/dev/null
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 JT_ADAM_EARLY=0 $B > gpurun_out/r6r_w12_noearly.json 2>/dev/null
JT_BWD_SPLIT=8 JT_ADAM_EARLY=0 $B > gpurun_out/r6r_w8_noearly.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6r_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print("%-28s %.4f  bwd %.3f (chain %.3f scatter %.3f)"%(f.split("/")[-1][4:-5], d["ms_per_step"], r["launch_ms"], r.get("launch_ms_chain") or 0, r.get("launch_ms_scatter") or 0))
    except Exception as e: print(f, "ERR", e)
PY
