#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "llff" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "llff" 2>&1 | tail -2
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc --config bat_llff_VM_MLP"
for i in 1 2 3; do
$B > gpurun_out/r6v_llff_default_$i.json 2>/dev/null
JT_SCATTER_WAVES=8 $B > gpurun_out/r6v_llff_w8_$i.json 2>/dev/null
done
$B --stage 3 > gpurun_out/r6v_llff3_default_1.json 2>/dev/null
JT_SCATTER_WAVES=8 $B --stage 3 > gpurun_out/r6v_llff3_w8_1.json 2>/dev/null
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r6v_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
    acc[f.split("/")[-1][4:-7]].append((d["ms_per_step"], r.get("launch_ms_scatter")))
for k,v in acc.items(): print(k, ["%.3f (%.3f)"%x for x in v])
PY
