#!/bin/bash
# round 6, run au: the whole bat_llff_VM_MLP schedule on a synthetic forward-facing scene, final code (a soak and an end-to-end time)
cd $GRAFT_REPO_ROOT
timeout 900 python tools/converge.py --config bat_llff_VM_MLP --compress 1 --image-size 0 --views 18 --graph 2>gpurun_out/r6au.err | grep '^{' > gpurun_out/round6_full_schedule_llff_final.jsonl
cut -c1-500 gpurun_out/round6_full_schedule_llff_final.jsonl | tail -16
tail -5 gpurun_out/r6au.err
