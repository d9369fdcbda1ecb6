"""What is the host doing inside the step's torch.rand?  Wall clock vs thread CPU time per call, a histogram over the steps, and the
same with the cyclic garbage collector off.  GPU box only."""
import os, sys, time, gc
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
if os.environ.get("PROBE_GC_OFF") == "1":
    gc.disable()
if os.environ.get("PROBE_GC_FREEZE") == "1":
    pass
W, C = [], []
orig = torch.rand


def rand(*a, **k):
    w, c = time.perf_counter(), time.thread_time()
    try:
        return orig(*a, **k)
    finally:
        W.append(time.perf_counter() - w)
        C.append(time.thread_time() - c)


torch.rand = rand
import threading
import bench
if os.environ.get("PROBE_GC_FREEZE") == "1":
    import joint_tensorf_amd  # noqa
    gc.collect(); gc.freeze()
bench.main()
import numpy as np
w, c = np.array(W[-300:]) * 1e6, np.array(C[-300:]) * 1e6
print("torch.rand in the step, last 300 calls: wall mean %.1f median %.1f p90 %.1f max %.1f us; thread CPU mean %.1f median %.1f us"
      % (w.mean(), np.median(w), np.percentile(w, 90), w.max(), c.mean(), np.median(c)))
print("threads:", [t.name for t in threading.enumerate()], "gc:", gc.isenabled(), gc.get_count(), gc.get_threshold())
print("first 40 of them (wall us):", " ".join("%.0f" % x for x in w[:40]))
