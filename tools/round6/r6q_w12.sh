#!/bin/bash
# round 6, run q: the 48-channel scatter at three waves per SIMD (twelve-wave workgroups, runs of 8) -- an experiment
cd $GRAFT_REPO_ROOT
timeout 300 env JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 python -m pytest tests/test_gpu_parity.py -x -q -k "blender_train_mid and (mfma-split8 or mfma-split16)" 2>&1 | tail -2
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
for i in 1 2; do
$B > gpurun_out/r6q_default_$i.json 2>/dev/null
JT_BWD_SPLIT=8 $B > gpurun_out/r6q_split8_w8_$i.json 2>/dev/null
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 $B > gpurun_out/r6q_split8_w12_$i.json 2>/dev/null
done
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 JT_NO_AUX=1 JT_ADAM_EARLY=0 JT_SCATTER_WGS=256 $B > gpurun_out/r6q_split8_w12_alone.json 2>/dev/null
JT_BWD_SPLIT=8 JT_NO_AUX=1 JT_ADAM_EARLY=0 JT_SCATTER_WGS=256 $B > gpurun_out/r6q_split8_w8_alone.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6q_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print("%-28s %.4f  bwd %.3f (chain %.3f scatter %.3f)"%(f.split("/")[-1][4:-5], d["ms_per_step"], r["launch_ms"], r.get("launch_ms_chain") or 0, r.get("launch_ms_scatter") or 0))
    except Exception as e: print(f, "ERR", e)
PY
