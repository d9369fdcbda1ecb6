#!/bin/bash
cd $GRAFT_REPO_ROOT
sleep 20   # an idle GPU, as the driver's fresh box
timeout 300 python tools/round6/clock_ramp.py --no-extras --no-cpu-baseline --no-torch-baseline --no-probe --no-live-pmc --steps 60 --warmup 0 2>&1 | grep -v '^{' | tail -4
