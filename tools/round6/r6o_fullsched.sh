#!/bin/bash
# the whole 40 000-iteration bat_blender_VM schedule on the self-consistent rendered scene (the reference's acceptance flow), round-6 code
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/converge.py --compress 1 --image-size 400 --views 100 --graph 2>&1 | grep '^{' > gpurun_out/round6_full_schedule_rendered_scene.jsonl
tail -n 3 gpurun_out/round6_full_schedule_rendered_scene.jsonl | cut -c1-1200
JT_LEAN_TAPE=0 timeout 1500 python tools/converge.py --compress 1 --image-size 400 --views 100 --graph 2>&1 | grep '^{' > gpurun_out/round6_full_schedule_rendered_scene_fulltape.jsonl
tail -n 2 gpurun_out/round6_full_schedule_rendered_scene_fulltape.jsonl | head -1 | cut -c1-400
