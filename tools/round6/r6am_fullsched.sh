#!/bin/bash
# round 6, run am: the whole 40 000-iteration bat_blender_VM schedule on the self-consistent rendered scene, the round's FINAL code
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/converge.py --compress 1 --image-size 400 --views 100 --graph 2>gpurun_out/r6am.err | grep '^{' > gpurun_out/round6_full_schedule_rendered_scene_final.jsonl
cat gpurun_out/round6_full_schedule_rendered_scene_final.jsonl | cut -c1-600 | tail -14
tail -5 gpurun_out/r6am.err
