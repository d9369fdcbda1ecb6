#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > $O/r4_fulltests.log 2>&1; echo "gpu suite rc=$? $(tail -1 $O/r4_fulltests.log)"
grep -E "FAILED|Error" $O/r4_fulltests.log | head -20
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
