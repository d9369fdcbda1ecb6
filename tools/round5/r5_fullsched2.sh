#!/bin/bash
# the whole 40 000-iteration bat_blender_VM schedule on the self-consistent rendered scene with the round's final code
# (stepper with the per-stage launch-mode choice; round 3: 104 s, round 4: 99.4 s)
cd $GRAFT_REPO_ROOT
python tools/converge.py --compress 40 --image-size 200 --views 40 --graph > /dev/null 2>&1   # warms the box
timeout 1500 python tools/converge.py --compress 1 --image-size 400 --views 100 --graph 2>&1 | grep '^{' > gpurun_out/round5_full_schedule_rendered_scene.jsonl
tail -n 3 gpurun_out/round5_full_schedule_rendered_scene.jsonl | cut -c1-1500
