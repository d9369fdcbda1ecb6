#!/bin/bash
# the secondary workloads with and without the round's last two changes (scatter on 192 CUs, early optimizer step)
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],3))"; }
for rep in 1 2; do
echo "configs3   default $($B --total-rays 65536 --steps 8 --warmup 2 2>/dev/null | get)   wgs256 $(JT_SCATTER_WGS=256 $B --total-rays 65536 --steps 8 --warmup 2 2>/dev/null | get)   wgs256+adam-late $(JT_SCATTER_WGS=256 JT_ADAM_EARLY=0 $B --total-rays 65536 --steps 8 --warmup 2 2>/dev/null | get)"
echo "stage4 blur default $($B --stage 4 --it 9000 --steps 40 2>/dev/null | get)   wgs256 $(JT_SCATTER_WGS=256 $B --stage 4 --it 9000 --steps 40 2>/dev/null | get)   wgs256+adam-late $(JT_SCATTER_WGS=256 JT_ADAM_EARLY=0 $B --stage 4 --it 9000 --steps 40 2>/dev/null | get)"
echo "stage 0     default $($B --stage 0 --steps 40 2>/dev/null | get)   adam-late $(JT_ADAM_EARLY=0 $B --stage 0 --steps 40 2>/dev/null | get)"
echo "fitted      default $($B --scene fitted --steps 40 2>/dev/null | get)   adam-late $(JT_ADAM_EARLY=0 $B --scene fitted --steps 40 2>/dev/null | get)"
done
