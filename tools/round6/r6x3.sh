#!/bin/bash
cd $GRAFT_REPO_ROOT
F="--scene fitted --no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe --steps 300 --warmup 20"
for rep in 1 2; do
for mt in 1 0; do
MT=$mt timeout 600 python -c "
import os, sys, torch
sys.argv = ['bench.py'] + '$F'.split()
if os.environ['MT'] == '0':
    torch.autograd.set_multithreading_enabled(False)
import bench
bench.main()
" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('multithreaded autograd=$mt ms_per_step', json.loads(l)['ms_per_step'])"
done; done
