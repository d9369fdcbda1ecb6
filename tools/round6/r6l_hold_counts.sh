#!/bin/bash
# round 6, run l: does the held-flush variant issue fewer vector-memory instructions?  SQ_INSTS_VMEM of k_shade_scatter, library
# as shipped against the held-flush variant (tools/round6/held_flush_experiment.patch built by tools/build_variant.py)
cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh r6l_default > gpurun_out/r6l_default.txt 2>&1
JT_LIB_PATH=$GRAFT_REPO_ROOT/joint_tensorf_amd/lib/variants/hold.so bash tools/pmc_sq.sh r6l_hold > gpurun_out/r6l_hold.txt 2>&1
grep -E 'kernel|k_shade_scatter|k_march_bwd_walk' gpurun_out/r6l_default.txt gpurun_out/r6l_hold.txt
