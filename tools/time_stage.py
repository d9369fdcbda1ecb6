#!/usr/bin/env python3
"""Time bench steps with torch.cuda events for a few env settings (profiling helper)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for abl in sys.argv[1:]:
    env = dict(os.environ, JT_ABLATE=abl)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stage", "4", "--steps", "10", "--warmup", "3",
                          "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    try:
        print("JT_ABLATE=%s ms_per_step=%.3f" % (abl, json.loads(out)["ms_per_step"]))
    except Exception as e:
        print("JT_ABLATE=%s failed: %s" % (abl, out[:300]))
