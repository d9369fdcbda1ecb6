#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4 or 299cube or configs3 or chunk" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_guards.py tests/test_gpu_graph.py tests/test_gpu_trajectory.py tests/test_engine_trace.py -x -q 2>&1 | tail -2
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
for i in 1 2 3; do
$B > gpurun_out/r6u_default_$i.json 2>/dev/null
JT_SCATTER_WAVES=8 $B > gpurun_out/r6u_w8_$i.json 2>/dev/null
JT_LEAN_TAPE=0 $B > gpurun_out/r6u_fulltape_$i.json 2>/dev/null
done
$B --n-voxel-final 27000000 --n-rays 4096 > gpurun_out/r6u_parent_1.json 2>/dev/null
JT_SCATTER_WAVES=8 $B --n-voxel-final 27000000 --n-rays 4096 > gpurun_out/r6u_parent_w8_1.json 2>/dev/null
$B --total-rays 65536 --steps 8 --warmup 2 > gpurun_out/r6u_cfg3_1.json 2>/dev/null
JT_SCATTER_WAVES=8 $B --total-rays 65536 --steps 8 --warmup 2 > gpurun_out/r6u_cfg3_w8_1.json 2>/dev/null
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r6u_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    acc[f.split("/")[-1][4:-7]].append(d["ms_per_step"])
for k,v in acc.items(): print(k, ["%.3f"%x for x in v], "median %.3f"%sorted(v)[len(v)//2])
PY
