"""Where the HOST time of a host-bound eager training step goes: wall-clock sections around the pieces of bench.py's step
(no profiler: cProfile inflates Python frames and does not see the autograd worker thread), SELF time per section (its own time
minus the sections nested in it) over the timed steps, and the timeline of one step.  GPU box only.
  python tools/round6/host_sections.py [bench.py flags]"""
import os, sys, time, collections, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
LOG = []          # (tag, start, end, thread, depth)
DEPTH = collections.defaultdict(int)
pc = time.perf_counter


def timed(f, tag):
    def g(*a, **k):
        th = threading.get_ident()
        d = DEPTH[th]
        DEPTH[th] = d + 1
        t = pc()
        try:
            return f(*a, **k)
        finally:
            LOG.append((tag, t, pc(), th, d))
            DEPTH[th] = d
    return g


def wrap(obj, name, tag):
    setattr(obj, name, timed(getattr(obj, name), tag))


class LibProxy:
    def __init__(self, lib):
        self.__dict__["_l"] = lib
        self.__dict__["_c"] = {}

    def __getattr__(self, n):
        c = self._c
        if n not in c:
            f = getattr(self._l, n)
            c[n] = timed(f, "C:" + n) if callable(f) else f
        return c[n]


for n in ("rand", "empty", "zeros", "empty_like", "zeros_like"):
    wrap(torch, n, "torch." + n)
wrap(torch.Tensor, "backward", "loss.backward")
import bench
from joint_tensorf_amd import ops, optim, tensorf_repr
from joint_tensorf_amd.model import bat_hip
proxy = LibProxy(ops.lib)
for m in (ops, optim, tensorf_repr, bat_hip):
    if hasattr(m, "lib"):
        m.lib = proxy
for cls, names in ((bat_hip.Graph, ("forward", "compute_loss", "get_pose", "render")),
                   (bat_hip.Model, ("summarize_loss", "after_iteration", "reduce_pose_gradients")),
                   (optim.VMAdam, ("step", "zero_grad")), (tensorf_repr.TensorVMSplit if hasattr(tensorf_repr, "TensorVMSplit") else bat_hip.Graph, ())):
    for n in names:
        wrap(cls, n, "%s.%s" % (cls.__name__, n))
wrap(ops, "render_rays", "ops.render_rays")
for n in ("_factors_struct", "_mlp_struct", "factor_storage", "_workspace", "status_word"):
    if hasattr(ops, n):
        wrap(ops, n, "ops." + n)
for n in dir(ops):
    o = getattr(ops, n)
    if isinstance(o, type) and issubclass(o, torch.autograd.Function) and o is not torch.autograd.Function:
        for m in ("forward", "backward"):
            if m in o.__dict__:
                f = o.__dict__[m]
                f = f.__func__ if isinstance(f, staticmethod) else f
                setattr(o, m, staticmethod(timed(f, "%s.%s" % (n, m))))
orig_one = None
STEP_MARK = []
_pc0 = bench.time.perf_counter


bench.main()
# the timed steps: every loss.backward of the main thread closes one
bw = [r for r in LOG if r[0] == "loss.backward"]
n_steps = min(200, len(bw) - 5)
t_first = bw[-n_steps - 1][2]
t_last = bw[-1][2]
rows = sorted((r for r in LOG if t_first <= r[1] and r[2] <= t_last + 1), key=lambda r: (r[3], r[1], -r[2]))
self_t = collections.defaultdict(float)
calls = collections.defaultdict(int)
stack = {}
for tag, s, e, th, d in rows:
    st = stack.setdefault(th, [])
    while st and st[-1][2] <= s:
        st.pop()
    if st:
        self_t[st[-1][0]] -= e - s
    self_t[tag] += e - s
    calls[tag] += 1
    st.append((tag, s, e))
span = (t_last - t_first) / n_steps
print("host sections over the last %d steps: %.1f us per step wall clock; SELF time per section (nested sections subtracted):" % (n_steps, span * 1e6))
acc = 0.0
for k, v in sorted(self_t.items(), key=lambda kv: -kv[1]):
    acc += v
    print("  %-34s %8.1f us  (%5.2f calls per step)" % (k, 1e6 * v / n_steps, calls[k] / n_steps))
print("  %-34s %8.1f us" % ("(outside every section)", 1e6 * (span - sum(v for k, v in self_t.items() if True and k) / n_steps)))
# timelines of three steps (main thread and the autograd thread)
main = threading.get_ident()
for k in (4, 3, 2):
    a, b = bw[-k - 1][2], bw[-k][2]
    print("timeline of a step (us from its start, duration; '|' = autograd thread; gaps > 25 us between consecutive entries marked):")
    prev_end = a
    for tag, s, e, th, d in sorted((r for r in LOG if a <= r[1] < b), key=lambda r: r[1]):
        if (tag.startswith("torch.") or tag.startswith("ops.")) and e - s < 4e-6:
            continue
        if s - prev_end > 25e-6:
            print("  %8s   ... %.0f us in Python / torch outside the sections" % ("", (s - prev_end) * 1e6))
        print("  %8.1f %s%s%-30s %7.1f" % ((s - a) * 1e6, "" if th == main else "| ", "  " * d, tag, (e - s) * 1e6))
        prev_end = max(prev_end, s) if d == 0 and e - s > 100e-6 else max(prev_end, e)
    if os.environ.get("ONE_TIMELINE") == "1":
        break
