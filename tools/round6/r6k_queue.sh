#!/bin/bash
# round 6, run k: the issuer wave (k_shade_scatter FLAGS bit 2): parity with a timeout guard, then alternating bench runs
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "blender_train_mid and mfma" 2>&1 | tail -3 > gpurun_out/r6k_parity.txt
if grep -q passed gpurun_out/r6k_parity.txt && ! grep -q failed gpurun_out/r6k_parity.txt; then
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 >> gpurun_out/r6k_parity.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp or llff_final_grid or configs3" 2>&1 | tail -3 >> gpurun_out/r6k_parity.txt
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
for i in 1 2 3; do
$B > gpurun_out/r6k_bench_queue_$i.json 2>/dev/null
JT_SCATTER_QUEUE=0 $B > gpurun_out/r6k_bench_noqueue_$i.json 2>/dev/null
done
JT_SCATTER_WGS=256 $B > gpurun_out/r6k_bench_queue_wg256.json 2>/dev/null
JT_SCATTER_WGS=192 $B > gpurun_out/r6k_bench_queue_wg192.json 2>/dev/null
JT_NO_AUX=1 JT_ADAM_EARLY=0 JT_SCATTER_WGS=256 $B > gpurun_out/r6k_bench_queue_alone.json 2>/dev/null
JT_SCATTER_QUEUE=0 JT_NO_AUX=1 JT_ADAM_EARLY=0 JT_SCATTER_WGS=256 $B > gpurun_out/r6k_bench_noqueue_alone.json 2>/dev/null
$B --config bat_llff_VM_MLP > gpurun_out/r6k_bench_llff_queue.json 2>/dev/null
JT_SCATTER_QUEUE=0 $B --config bat_llff_VM_MLP > gpurun_out/r6k_bench_llff_noqueue.json 2>/dev/null
fi
cat gpurun_out/r6k_parity.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6k_bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print("%-28s %.4f  bwd %.3f (chain %.3f scatter %.3f)"%(f.split("/")[-1][10:-5], d["ms_per_step"], r["launch_ms"], r.get("launch_ms_chain") or 0, r.get("launch_ms_scatter") or 0))
    except Exception as e: print(f, "ERR", e)
PY
