#!/bin/bash
# round 6, run n: test-time pose optimisation with the pose composed by the pose kernel on the cached aligned base
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_lifecycle.py tests/test_gpu_edge.py tests/test_gpu_fused.py -x -q 2>&1 | tail -4 > gpurun_out/r6n_eval.txt
timeout 600 python tools/eval_bench.py --graph --no-render 2>/dev/null | tail -1 > gpurun_out/r6n_evalbench_graph.json
timeout 600 python tools/eval_bench.py --no-render 2>/dev/null | tail -1 > gpurun_out/r6n_evalbench_eager.json
timeout 600 python tools/eval_bench.py --no-render --test-iters 40 --batch-views 1,8,32 2>/dev/null | tail -1 > gpurun_out/r6n_evalbench_batched.json
timeout 600 python tools/eval_bench.py --graph --no-render --scene blobs 2>/dev/null | tail -1 > gpurun_out/r6n_evalbench_graph_blobs.json
cat gpurun_out/r6n_eval.txt
python - <<'PY'
import json
for n in ("graph","eager","batched","graph_blobs"):
    try:
        d=json.loads(open("gpurun_out/r6n_evalbench_%s.json"%n).read())
        print(n, d.get("test_time_optim"), d.get("test_time_optim_batched_ms_per_view_iteration"))
    except Exception as e: print(n,"ERR",e)
PY
