#!/bin/bash
# round 6, run aq: is the Blender march occupancy-bound?  k_march_fwd / k_march_bwd_scan per ray at 1 995 and at 3 840 rays per iteration
cd $GRAFT_REPO_ROOT
NO_PMC=1 bash tools/profile_cmd.sh r6aq_r2048 > gpurun_out/r6aq_a.log 2>&1
NO_PMC=1 bash tools/profile_cmd.sh r6aq_r4096 --n-rays 4096 > gpurun_out/r6aq_b.log 2>&1
for t in r2048 r4096; do echo $t; grep -i 'march_fwd\|bwd_scan\|shade_fwd' gpurun_out/r6aq_${t}_trace_summary.txt; python -c "
import json
j=json.loads([l for l in open('gpurun_out/r6aq_${t}_bench_line.json') if l.startswith('{')][-1]); print(j['config']['rays_per_iter_per_gpu'], j['ms_per_step'])"; done
