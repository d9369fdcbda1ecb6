#!/usr/bin/env python3
"""Which host line issues each small launch of an eager training iteration?  torch.profiler over a few iterations of the default
bench workload with Python stacks; prints, for every kernel / memcpy shorter than 20 us, its count per iteration and the innermost
frames inside joint_tensorf_amd / bench.py that were active when it was launched.

  python tools/launch_sources.py [--config bat_blender_VM] [--steps 4]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from joint_tensorf_amd.options import Opt, make_options  # noqa: E402
from joint_tensorf_amd.synthetic import make_views  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="bat_blender_VM")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    torch.manual_seed(0)
    np.random.seed(0)
    opt = make_options(args.config, device="cuda:0")
    stage, it0 = bench.stage_setup(opt, 4)
    n_views = 100 if args.config == "bat_blender_VM" else 18
    model = bench.build_model(opt, it0, n_views)
    var_all = make_views(opt, n_views, seed=0, device="cuda:0")

    def step():
        model.before_iteration(opt)
        model.train_iteration(opt, Opt(dict(var_all)))
        model.after_iteration(opt)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    ev = prof.events()
    by_corr = {}
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CPU and getattr(e, "stack", None):
            for k in getattr(e, "kernels", []) or []:
                by_corr.setdefault(k.name, []).append((k.duration, e.stack, e.name))
    rows = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CPU:
            ks = getattr(e, "kernels", []) or []
            if not ks:
                continue
            frames = [f for f in (e.stack or []) if ("joint_tensorf_amd" in f or "bench.py" in f)]
            where = " <- ".join(os.path.basename(f.split(",")[0].replace("(", ":").strip()) if False else f.strip()[-70:] for f in frames[:2]) or e.name
            for k in ks:
                r = rows[k.name[:60]]
                r[0] += 1
                r[1] += k.duration
                r[2][where] += 1
    # the stock elementwise ops (add / mul / copy_ / fill_ ...) once more, with the Python line that called them
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    seen = collections.Counter()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if any(k in name for k in ("add", "mul", "copy_", "fill_", "zero_", "_to_copy", "clone", "arange", "sum", "index")):
                if any(torch.is_tensor(a) and a.is_cuda for a in args):
                    fr = [f for f in traceback.extract_stack() if ("joint_tensorf_amd" in f.filename or "bench.py" in f.filename)]
                    where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
                    seen[(name, where)] += 1
            return func(*args, **(kwargs or {}))
    with Log():
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize()
    print("stock elementwise ops on GPU tensors per iteration (forward pass and optimizer; autograd's own C++ calls not included):")
    for (name, where), c in seen.most_common(40):
        print("   %5.2f  %-28s %s" % (c / args.steps, name, where))
    print("%-60s %6s %8s  issued from" % ("launch", "/iter", "avg us"))
    for name, (n, dur, wh) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
        if dur / n > 20:
            continue
        print("%-60s %6.2f %8.1f" % (name, n / args.steps, dur / n))
        for w, c in wh.most_common(3):
            print("        %5.2f  %s" % (c / args.steps, w))


if __name__ == "__main__":
    main()
