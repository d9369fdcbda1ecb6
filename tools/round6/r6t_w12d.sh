#!/bin/bash
cd $GRAFT_REPO_ROOT
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mfma and not fp32 and not tile and not split16 and not fulltape" 2>&1 | tail -2
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp_400cube and mfma and not fp32 and not tile and not split16 and not fulltape" 2>&1 | tail -2
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
for i in 1 2 3 4; do
$B > gpurun_out/r6t_default_$i.json 2>/dev/null
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 JT_SCATTER_WGS=192 $B > gpurun_out/r6t_w12_wg192_$i.json 2>/dev/null
JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12 JT_SCATTER_WGS=160 $B > gpurun_out/r6t_w12_wg160_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r6t_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    acc[f.split("/")[-1][4:-7]].append(d["ms_per_step"])
for k,v in acc.items(): print(k, ["%.3f"%x for x in v], "median %.3f"%sorted(v)[len(v)//2])
PY
