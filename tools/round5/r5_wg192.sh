#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py -x -q 2>&1 | tail -n 3
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc"
get() { python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(round(j['ms_per_step'],3), 'chain', round(r.get('launch_ms_chain',0),3), 'scatter', round(r.get('launch_ms_scatter',0),3))"; }
for rep in 1 2; do
echo "default: $($B 2>/dev/null | get)"
echo "JT_SCATTER_WGS=256: $(JT_SCATTER_WGS=256 $B 2>/dev/null | get)"
done
echo "llff: $($B --config bat_llff_VM_MLP --it 30000 2>/dev/null | get)"
echo "graph: $(JT_GRAPH=1 $B 2>/dev/null | get)"
bash tools/round5/r5_timeline.sh > /dev/null 2>&1
cat gpurun_out/r5_timeline.txt
