#!/bin/bash
cd $GRAFT_REPO_ROOT
JT_WALK_DEBUG=1 JT_TIME_WALK=1 python3 - <<'PY' > gpurun_out/r5_walkdbg.txt 2>&1
import sys, atexit, torch
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-probe", "--no-torch-baseline", "--no-extras",
            "--config", "bat_llff_VM_MLP", "--no-live-pmc"]
from joint_tensorf_amd import ops
def dump():
    torch.cuda.synchronize()
    names = "R gflat g_o rays_o jitter zvals sigma_feat weight g_xyz mws".split()
    for t0, t1, *rest in ops._WALK_DBG:
        print("%7.1f us " % (t0.elapsed_time(t1) * 1e3) + " ".join("%s=%x" % (n, v) if n != "R" else "R=%d" % v for n, v in zip(names, rest)))
atexit.register(dump)
import runpy
runpy.run_path("bench.py", run_name="__main__")
PY
tail -45 gpurun_out/r5_walkdbg.txt
