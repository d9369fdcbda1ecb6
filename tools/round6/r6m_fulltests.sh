#!/bin/bash
# the whole GPU suite as the driver runs it, on the round's code; then smoke()
cd $GRAFT_REPO_ROOT
( time timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=8 ) > gpurun_out/r6m_fulltests.log 2>&1
tail -n 16 gpurun_out/r6m_fulltests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
