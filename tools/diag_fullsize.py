#!/usr/bin/env python3
"""Where do the largest gradient differences of the full-size parity cases come from?  Runs one case through the
product path, the fp32 oracle (twice: its own run-to-run noise) and the fp64 oracle, and locates the outliers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import fullsize_util as U  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    opt, model, var, it0 = U.build("bat_blender_VM", stage=-1, density_scale=scale or None)
    hip = U.run_hip(opt, model, var)
    hip2 = U.run_hip(opt, model, var)
    r32 = U.run_oracle(opt, model, var, hip["ctx"])
    r32b = U.run_oracle(opt, model, var, hip["ctx"])
    r64 = U.run_oracle(opt, model, var, hip["ctx"], dtype=torch.float64)
    U.compare("diag scale %g" % scale, opt, model, hip, r32, r64)
    rB = U.run_oracle(opt, model, var, hip["ctx"], linear_dtype=torch.float64)
    print("fp32 oracle vs fp32 oracle with fp64-accumulated Linear layers | hip vs either:")
    for k in r32["grads"]:
        print("   %-18s A-vs-B %.2e / %.2e   hip-vs-A %.2e / %.2e   hip-vs-B %.2e / %.2e" % (
            k, U.rel_max(r32["grads"][k], rB["grads"][k]), U.rel_l2(r32["grads"][k], rB["grads"][k]),
            U.rel_max(hip["grads"][k], r32["grads"][k]), U.rel_l2(hip["grads"][k], r32["grads"][k]),
            U.rel_max(hip["grads"][k], rB["grads"][k]), U.rel_l2(hip["grads"][k], rB["grads"][k])))
    print("run-to-run (same inputs):")
    for k in r32["grads"]:
        print("   %-18s hip %.2e   fp32 oracle %.2e" % (k, U.rel_max(hip["grads"][k], hip2["grads"][k]),
                                                          U.rel_max(r32["grads"][k], r32b["grads"][k])))
    for name in ("app_plane.0", "app_line.0"):
        gh, g32, g64 = hip["grads"][name].double(), r32["grads"][name].double(), r64["grads"][name]
        d = (gh - g64).abs()
        mx = g64.abs().max()
        print("%s: max |g64| %.3e; hip-vs-64 error quantiles (rel to max):" % (name, mx))
        q = torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999, 1.0], device=d.device, dtype=torch.float64)
        flat = (d / mx).flatten()
        sel = flat[torch.randperm(flat.numel(), device=d.device)[:4_000_000]]
        print("   hip   ", ["%.1e" % v for v in torch.quantile(sel, q).tolist()])
        flat2 = ((g32 - g64).abs() / mx).flatten()
        sel2 = flat2[torch.randperm(flat2.numel(), device=d.device)[:4_000_000]]
        print("   fp32  ", ["%.1e" % v for v in torch.quantile(sel2, q).tolist()])
        top = torch.topk(d.flatten(), 8).indices
        shp = gh.shape
        for t in top.tolist():
            idx = []
            r = t
            for s in reversed(shp):
                idx.append(r % s)
                r //= s
            idx = tuple(reversed(idx))
            print("   at %s: hip %.6e  fp32 %.6e  fp64 %.6e" % (idx, gh[idx], g32[idx], g64[idx]))
    print("shaded fp32 %d fp64 %d" % (r32["shaded"], r64["shaded"]))


if __name__ == "__main__":
    main()
