#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_eval.py tests/test_abi.py -x -q 2>&1 | tail -n 4 > gpurun_out/r5_bench_full_tests.log
( time python bench.py ) > gpurun_out/r5_bench_full_line.json 2> gpurun_out/r5_bench_full.err
tail -n 3 gpurun_out/r5_bench_full_tests.log
tail -n 5 gpurun_out/r5_bench_full.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r5_bench_full_line.json') if l.startswith('{')][-1])
r=j['roofline']
print(j['value'], j['ms_per_step'], r['launch_ms'], r['frac'], r.get('traffic'), r.get('traffic_source'))
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'], j['cpu_baseline']['sample'][:200])
for k,v in j.get('extra',{}).items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','rays_per_s','k_shade_bwd_ms','error','test_time_pose_optim_ms_per_iter','ms_per_image')})
PY
