#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_trace.py -q -s -m gpu 2>&1 | grep 'engine trace\]\|passed\|failed' > gpurun_out/r6h.txt
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_trajectory.py tests/test_gpu_dist.py::test_rccl_process_group_of_one_rank -x -q 2>&1 | tail -4 >> gpurun_out/r6h.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
$B > gpurun_out/r6h_bench_lean.json 2>/dev/null
JT_ADAM_EARLY=0 $B > gpurun_out/r6h_bench_lean_noearly.json 2>/dev/null
cat gpurun_out/r6h.txt
python - <<'PY'
import json
for n in ("lean","lean_noearly"):
    d=json.loads([l for l in open("gpurun_out/r6h_bench_%s.json"%n) if l.startswith("{")][-1])
    print(n, d["ms_per_step"], d.get("gpu_clocks_after_timed_loop"))
PY
