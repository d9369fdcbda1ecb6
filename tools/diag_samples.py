#!/usr/bin/env python3
"""Per-sample comparison of the march stage (density feature, weight, d loss / d density feature) between the product
path and the fp32 / fp64 oracle on one slice of the full-size stage-4 batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from joint_tensorf_amd import ops  # noqa: E402
from oracle import tensorf_oracle as O  # noqa: E402
from tests import fullsize_util as U  # noqa: E402


def q(x):
    x = x.flatten().double()
    if x.numel() > 4_000_000:
        x = x[torch.randperm(x.numel(), device=x.device)[:4_000_000]]
    qs = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=x.device, dtype=torch.float64)
    return " ".join("%.1e" % v for v in torch.quantile(x, qs).tolist())


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    ops.KEEP_INTERMEDIATES = True
    opt, model, var, it0 = U.build("bat_blender_VM", stage=-1, density_scale=scale or None)
    keep = {}
    orig_rg = ops.ray_gen

    def rg(*a, **k):
        c, r = orig_rg(*a, **k)
        if c.requires_grad:
            c.retain_grad()
            keep["c"] = c
        return c, r
    ops.ray_gen = rg
    hip = U.run_hip(opt, model, var)
    ops.ray_gen = orig_rg
    gc_h = keep["c"].grad.clone()
    tf = model.graph.nerf.tensorf
    ctx = hip["ctx"]
    R, S, B, n_lat = ctx["R"], ctx["S"], ctx["B"], ctx["n_lat"]
    inter = tf.last_render_cfg.intermediates
    w_h, f_h = inter["weight"].view(B, n_lat, S), inter["sigma_feat"].view(B, n_lat, S)
    gfeat_h = ops._WS[(str(w_h.device), "march_bwd")][:R * S * 4].view(torch.float32).view(B, n_lat, S).clone()
    kept = {}
    orig = O.render

    def orender(*a, **k):
        k["return_aux"] = True
        a[2].retain_grad()
        kept["c"] = a[2]
        out = orig(*a, **k)
        aux = out[3]
        aux["sigma"].retain_grad()
        kept.setdefault("aux", []).append(aux)
        return out
    O.render = orender
    try:
        for dt in (torch.float32, torch.float64):
            kept.clear()
            U.run_oracle(opt, model, var, ctx, dtype=dt, slice_rays=10 ** 9)
            aux = kept["aux"][0]
            w_o = aux["weight"].detach().view(B, n_lat, S)
            valid = aux["valid"].view(B, n_lat, S)
            f_o = torch.zeros(B, n_lat, S, device=w_o.device, dtype=dt)
            f_o[valid] = aux["sigma_feat"].detach()
            print("== oracle %s" % dt)
            print("  |feat_hip - feat| / max|feat| quantiles (50 90 99 99.9 100): %s" % q((f_h - f_o).abs()[valid] / f_o.abs().max()))
            print("  |w_hip - w| / w (w > thres): %s" % q(((w_h - w_o).abs() / w_o.abs().clamp_min(1e-30))[w_o > 1e-6]))
            gs = aux["sigma"].grad.view(B, n_lat, S)  # d loss / d sigma
            # d loss / d feat = d loss / d sigma * act'(feat + shift)
            x = f_o + float(opt.arch.density_shift)
            act = torch.sigmoid(x) if opt.arch.feature_to_density_activation == "softplus" else (x > 0).to(dt)
            gfeat_o = gs * act * valid
            print("  |gfeat_hip - gfeat| / max|gfeat|: %s   (max |gfeat| %.3e)" % (q((gfeat_h - gfeat_o).abs()[valid] / gfeat_o.abs().max()), float(gfeat_o.abs().max())))
            print("  |gfeat_hip - gfeat| / |gfeat| (|gfeat| > 1e-3 max): %s" % q(((gfeat_h - gfeat_o).abs() / gfeat_o.abs().clamp_min(1e-30))[gfeat_o.abs() > 1e-3 * gfeat_o.abs().max()]))
            gc_o = kept["c"].grad.view(B, n_lat, 3)
            err_r = (gc_h - gc_o).abs().amax(-1) / gc_o.abs().max()
            flip_r = (((w_h - w_o).abs() / w_o.abs().clamp_min(1e-30)) * (w_o > 1e-6) > 1e-3).any(-1)
            print("  per-ray g_center error / max: rays with an alpha-quantum flip (%d of %d): mean %.2e max %.2e; rays without: mean %.2e max %.2e" % (
                int(flip_r.sum()), flip_r.numel(), float(err_r[flip_r].mean()) if flip_r.any() else 0.0,
                float(err_r[flip_r].max()) if flip_r.any() else 0.0, float(err_r[~flip_r].mean()), float(err_r[~flip_r].max())))
            if dt == torch.float32:
                o32 = dict(w=w_o, gfeat=gfeat_o)
            else:
                print("  fp32 oracle vs fp64: |w32 - w| / w: %s" % q(((o32["w"] - w_o).abs() / w_o.abs().clamp_min(1e-30))[w_o > 1e-6]))
                print("  fp32 oracle vs fp64: |gfeat32 - gfeat| / |gfeat|: %s" % q(((o32["gfeat"] - gfeat_o).abs() / gfeat_o.abs().clamp_min(1e-30))[gfeat_o.abs() > 1e-3 * gfeat_o.abs().max()]))
    finally:
        O.render = orig


if __name__ == "__main__":
    main()
