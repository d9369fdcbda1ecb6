#!/bin/bash
# eager vs hipGraph-replayed train step per grid stage of bat_blender_VM (blurred stages included): DESIGN.md section 3
F="--no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-roofline --steps 40 --warmup 10"
for st in 0 1 2 3; do
  for g in 0 1; do
    JT_GRAPH=$g python bench.py $F --stage $st 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('stage $st graph $g', round(j['ms_per_step'],3), 'ms', round(j['value']), 'rays/s', j['config']['launch'])"
  done
done
for g in 0 1; do JT_GRAPH=$g python bench.py $F --stage 4 --it 9000 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('stage 4 blurred graph $g', round(j['ms_per_step'],3), 'ms', round(j['value']), 'rays/s', j['config']['launch'])"; done
