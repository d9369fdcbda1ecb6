#!/bin/bash
cd $GRAFT_REPO_ROOT
F="--scene fitted --no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe --steps 300 --warmup 20"
for m in none PROBE_GC_OFF PROBE_GC_FREEZE; do
  echo "== $m"
  env $m=1 timeout 300 python tools/round6/host_rand_probe.py $F 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('ms_per_step', json.loads(l)['ms_per_step'])
    else: print(l.rstrip()[:600])"
done
