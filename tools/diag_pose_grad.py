#!/usr/bin/env python3
"""Where does the se3-gradient difference between the product path and the fp32 oracle arise: per-ray gradients
(d loss / d origin, direction), per-view pose gradients, or the se3 chain?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from joint_tensorf_amd import ops  # noqa: E402
from oracle import tensorf_oracle as O  # noqa: E402
from tests import fullsize_util as U  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    opt, model, var, it0 = U.build("bat_blender_VM", stage=-1, density_scale=scale or None)
    kept = {}
    orig_rg, orig_tp = ops.ray_gen, ops.train_pose

    def rg(*a, **k):
        c, r = orig_rg(*a, **k)
        if c.requires_grad:
            c.retain_grad(); r.retain_grad()
            kept["c"], kept["r"] = c, r
        return c, r

    def tp(*a, **k):
        p = orig_tp(*a, **k)
        if p.requires_grad:
            p.retain_grad()
            kept["pose"] = p
        return p
    ops.ray_gen, ops.train_pose = rg, tp
    try:
        hip = U.run_hip(opt, model, var)
    finally:
        ops.ray_gen, ops.train_pose = orig_rg, orig_tp
    h = dict(gc=kept["c"].grad.clone(), gr=kept["r"].grad.clone(), gpose=kept["pose"].grad.clone(), se3=hip["grads"]["se3"])

    okept = {}
    o_rp, o_tp = O.rays_for_pixels, O.train_pose

    def orp(*a, **k):
        c, r = o_rp(*a, **k)
        okept.setdefault("cr", []).append((c, r))
        return c, r

    def otp(*a, **k):
        p = o_tp(*a, **k)
        p.retain_grad()
        okept.setdefault("pose", []).append(p)
        return p
    O.rays_for_pixels, O.train_pose = orp, otp
    # pinned rays are c + (c_hip - c).detach(): the gradient of the pinned tensor equals the gradient w.r.t. c
    orig_render = O.render

    def orender(cfg, params, c, r, *a, **k):
        c.retain_grad(); r.retain_grad()
        okept.setdefault("flat", []).append((c, r))
        return orig_render(cfg, params, c, r, *a, **k)
    O.render = orender
    try:
        for dt in (torch.float32, torch.float64):
            okept.clear()
            ref = U.run_oracle(opt, model, var, hip["ctx"], dtype=dt)
            B, n_lat = hip["ctx"]["B"], hip["ctx"]["n_lat"]
            gc = torch.cat([c.grad.view(B, -1, 3) for c, r in okept["flat"]], 1)
            gr = torch.cat([r.grad.view(B, -1, 3) for c, r in okept["flat"]], 1)
            gpose = sum(p.grad for p in okept["pose"])
            print("oracle %s vs hip:   g_center %.2e / %.2e   g_ray %.2e / %.2e   g_pose %.2e / %.2e   se3 %.2e / %.2e" % (
                dt, U.rel_max(h["gc"], gc), U.rel_l2(h["gc"], gc), U.rel_max(h["gr"], gr), U.rel_l2(h["gr"], gr),
                U.rel_max(h["gpose"], gpose), U.rel_l2(h["gpose"], gpose), U.rel_max(h["se3"], ref["grads"]["se3"]),
                U.rel_l2(h["se3"], ref["grads"]["se3"])))
            if dt == torch.float32:
                o32 = dict(gc=gc, gr=gr, gpose=gpose, se3=ref["grads"]["se3"])
            else:
                print("oracle fp32 vs fp64: g_center %.2e / %.2e   g_ray %.2e / %.2e   g_pose %.2e / %.2e   se3 %.2e / %.2e" % (
                    U.rel_max(o32["gc"], gc), U.rel_l2(o32["gc"], gc), U.rel_max(o32["gr"], gr), U.rel_l2(o32["gr"], gr),
                    U.rel_max(o32["gpose"], gpose), U.rel_l2(o32["gpose"], gpose), U.rel_max(o32["se3"], ref["grads"]["se3"]),
                    U.rel_l2(o32["se3"], ref["grads"]["se3"])))
                # the pose chain alone: feed the HIP per-ray gradients through the fp64 oracle's ray generator
                print("g_pose magnitude: max %.3e; per-ray g_center max %.3e (sum of |.| over rays of a view: %.3e)" % (
                    float(gpose.abs().max()), float(gc.abs().max()), float(gc.abs().sum(1).max())))
    finally:
        O.rays_for_pixels, O.train_pose, O.render = o_rp, o_tp, orig_render


if __name__ == "__main__":
    main()
