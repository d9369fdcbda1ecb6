#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --total-rays 65536 --steps 6 --warmup 2 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras"
echo default; $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['launch_ms'])"
echo split0_noaux; JT_BWD_SPLIT=0 JT_NO_AUX=1 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['launch_ms'])"
echo split16_noaux; JT_NO_AUX=1 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['launch_ms'])"
echo split0_aux; JT_BWD_SPLIT=0 JT_NO_AUX=0 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['launch_ms'])"
