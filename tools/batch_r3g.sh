#!/bin/bash
# round-3 evidence batch: new full-size cases, the default bench line, rocprofv3 traces + FETCH / WRITE passes of the default,
# LLFF-final-grid and fitted-scene steps, the roctx marker trace, the whole 40 000-iteration schedule on the rendered scene
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "middle or unpinned" 2>&1 | grep -v amdgpu.ids | tail -120) > gpurun_out/r3g_fullsize_new.log
(timeout 600 python bench.py > gpurun_out/r3g_bench_default.json 2> gpurun_out/r3g_bench_default.err)
timeout 600 tools/profile_cmd.sh r3g_default > gpurun_out/r3g_default_prof.log 2>&1
timeout 600 tools/profile_cmd.sh r3g_llff --config bat_llff_VM_MLP > gpurun_out/r3g_llff_prof.log 2>&1
timeout 600 tools/profile_cmd.sh r3g_fitted --scene fitted > gpurun_out/r3g_fitted_prof.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --marker-trace --stats --output-format csv -d $R/gpurun_out/r3g_marker -o m -- python3 $R/tools/marker_run.py > $R/gpurun_out/r3g_marker.log 2>&1)
(timeout 900 python tools/converge.py --compress 1 --image-size 400 --views 100 --graph --report-every 2000 2>&1 | grep "^{") > gpurun_out/r3g_full_schedule_rendered.jsonl
tail -3 gpurun_out/r3g_fullsize_new.log; tail -c 400 gpurun_out/r3g_bench_default.json; ls gpurun_out/r3g_marker; tail -3 gpurun_out/r3g_full_schedule_rendered.jsonl
