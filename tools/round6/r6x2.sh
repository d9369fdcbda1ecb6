#!/bin/bash
cd $GRAFT_REPO_ROOT
F="--scene fitted --no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
timeout 600 python tools/round6/host_backward_probe.py $F --steps 60 --warmup 20 2>&1 | grep -v '^{' | tail -80
