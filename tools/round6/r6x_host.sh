#!/bin/bash
# round 6, run x: host sections of the host-bound eager step (fitted scene)
cd $GRAFT_REPO_ROOT
F="--scene fitted --no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
timeout 600 python -m pytest tests/test_gpu_units.py -x -q -k "vmadam or adam" 2>&1 | tail -3
for v in 0 1; do JT_ADAM_PLAN=$v timeout 600 python bench.py $F --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('JT_ADAM_PLAN=$v ms_per_step', json.loads(l)['ms_per_step'])"; done
ONE_TIMELINE=1 timeout 600 python tools/round6/host_sections.py $F --steps 300 --warmup 20 2>&1 | grep -v '^{' | grep -v factor_storage | tail -150 > gpurun_out/r6x_sections.txt
timeout 600 python tools/round6/host_cprofile.py $F --steps 300 --warmup 20 2>&1 | grep -v '^{' > gpurun_out/r6x_cprofile.txt
tail -5 gpurun_out/r6x_cprofile.txt
