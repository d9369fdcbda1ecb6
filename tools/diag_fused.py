#!/usr/bin/env python3
"""Diagnostics of the single-launch test-time kernel (csrc/jt_fused.hip): (1) per-ray d loss / d (o, d) against the staged
kernels on a small scene, worst rays listed; (2) launch time of the fused op alone at the bench size, and of its phases
through JT_FUSED_ABLATE child processes.  usage: python tools/diag_fused.py [--time] [--scene blobs]"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rays_and_model(config, B, hw, grid, n_rays, dens, scene="random", size=None):
    from joint_tensorf_amd import ops
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import bake_blobs, make_views
    over = dict(data=dict(image_size=list(hw), num_views=B), nerf=dict(n_rays=n_rays))
    if grid:
        over["train_schedule"] = dict(n_voxel_init=grid, n_rays_init=n_rays, n_rays_rest=n_rays)
    opt = make_options(config, device="cuda", **over)
    if not grid:
        import bench
        bench.stage_setup(opt, -1)
    torch.manual_seed(0)
    np.random.seed(0)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=B)
    g = model.graph
    tf = g.nerf.tensorf
    with torch.no_grad():
        if scene == "blobs":
            bake_blobs(tf, n_blobs=12, seed=0)
        elif dens:
            for p in tf.density_plane:
                p.mul_(dens)
    g.nerf.set_progress(1.0)
    var = make_views(opt, B, seed=5, device="cuda")
    step = g.lattice_step(opt, B)
    sx = torch.arange(step // 2, opt.W, step, device="cuda")
    sy = torch.arange(step // 3, opt.H, step, device="cuda")
    ray_idx = (sx[None, :] + sy[:, None] * opt.W).reshape(-1)
    with torch.no_grad():
        c, r = ops.ray_gen(var.pose, var.intr_inv, var.intr, ray_idx, opt.W, ndc=bool(opt.camera.ndc))
    for p in g.parameters():
        p.requires_grad_(False)
    return opt, model, var, ray_idx, c, r


def both_paths(opt, model, var, ray_idx, c, r):
    from joint_tensorf_amd import ops
    g, tf = model.graph, model.graph.nerf.tensorf
    B, n = c.shape[0], c.shape[1]
    out = {}
    for name in ("staged", "fused"):
        o = c.reshape(-1, 3).clone().requires_grad_(True)
        d = r.reshape(-1, 3).clone().requires_grad_(True)
        if name == "fused":
            loss, rgb, depth, acc = tf.render_pose_fused(opt, o, d, var.image, ray_idx, n, white_bg=opt.nerf.setbg_opaque,
                                                         ndc_ray=opt.camera.ndc, N_samples=g.nerf.n_samples)
        else:
            rgb, depth, acc = tf.forward(opt, o, d, white_bg=opt.nerf.setbg_opaque, is_train=False, ndc_ray=opt.camera.ndc,
                                         N_samples=g.nerf.n_samples)
            loss = ops.render_loss(rgb.view(B, n, 3), var.image, ray_idx)
        loss.backward()
        out[name] = dict(loss=float(loss), rgb=rgb.detach(), go=o.grad.clone(), gd=d.grad.clone(), acc=acc.detach())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--scene", default="random")
    ap.add_argument("--config", default="bat_blender_VM")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    if not args.time:
        llff = args.config != "bat_blender_VM"
        opt, model, var, ray_idx, c, r = rays_and_model(args.config, 1, (36, 48) if llff else (48, 48), 9000 if llff else 20 ** 3,
                                                        300, 1.0 if llff else 22.0)
        out = both_paths(opt, model, var, ray_idx, c, r)
        a, b = out["staged"], out["fused"]
        keep = os.environ.get("JT_FUSED_ABLATE", "0")
        os.environ["JT_FUSED_ABLATE"] = "4"
        nb = both_paths(opt, model, var, ray_idx, c, r)["fused"]   # without the appearance backward
        os.environ["JT_FUSED_ABLATE"] = keep
        for k in ("go", "gd"):
            app = b[k] - nb[k]
            print(k, "appearance part of the fused gradient: max per axis", app.abs().max(0).values.tolist())
            print(k, "fused - staged: max per axis", (b[k] - a[k]).abs().max(0).values.tolist())
            print(k, "fused(no app) - staged: max per axis", (nb[k] - a[k]).abs().max(0).values.tolist())
        print("loss", a["loss"], b["loss"], "rgb max diff", float((a["rgb"] - b["rgb"]).abs().max()))
        # the arbiter: the oracle's stock torch ops, float64
        from oracle import tensorf_oracle as O
        tf, g = model.graph.nerf.tensorf, model.graph
        sd = {k: v.detach().cpu().double().contiguous() for k, v in tf.state_dict().items()}
        params = O.params_from_state_dict(sd, prefix="")
        cfg = O.SceneCfg(opt.data.scene_bbox, tf.gridSize.tolist(), [float(tf.near_far[0]), float(tf.near_far[1])],
                         step_ratio=opt.nerf.step_ratio)
        for kk in ("aabb", "aabbSize", "invaabbSize", "units", "stepSize"):
            setattr(cfg, kk, getattr(cfg, kk).double())
        oo = c.reshape(-1, 3).cpu().double().requires_grad_(True)
        dd = r.reshape(-1, 3).cpu().double().requires_grad_(True)
        rgb, _, _ = O.render(cfg, params, oo, dd, g.nerf.n_samples, white_bg=True)
        img = var.image.cpu().double().view(1, 3, -1).permute(0, 2, 1)
        l = O.render_loss(rgb.view(1, -1, 3), img[:, ray_idx.cpu()])
        l.backward()
        ref = dict(go=oo.grad.float().cuda(), gd=dd.grad.float().cuda())
        for k in ("go", "gd"):
            print(k, "vs fp64 oracle: staged rel l2 %.3e max %.3e | fused rel l2 %.3e max %.3e" % (
                float((a[k] - ref[k]).norm() / ref[k].norm()), float((a[k] - ref[k]).abs().max()),
                float((b[k] - ref[k]).norm() / ref[k].norm()), float((b[k] - ref[k]).abs().max())))
            for ax in range(3):
                print("    axis %d: staged max %.3e fused max %.3e" % (ax, float((a[k] - ref[k])[:, ax].abs().max()),
                                                                       float((b[k] - ref[k])[:, ax].abs().max())))
        for k in ("go", "gd"):
            d = (a[k] - b[k]).abs()
            print(k, "max |staged|", float(a[k].abs().max()), "max diff", float(d.max()), "rel l2",
                  float((a[k] - b[k]).norm() / a[k].norm()))
            worst = torch.topk(d.max(dim=1).values, 5).indices.tolist()
            for w in worst:
                print("   ray", w, "acc %.4f" % float(a["acc"][w]), "staged", [round(v, 9) for v in a[k][w].tolist()], "fused",
                      [round(v, 9) for v in b[k][w].tolist()])
        return
    # ---- timing at the bench size ----
    opt, model, var, ray_idx, c, r = rays_and_model(args.config, 1, (400, 400) if args.config == "bat_blender_VM" else (480, 640),
                                                    None, 2048 if args.config == "bat_blender_VM" else 4096, None, scene=args.scene)
    from joint_tensorf_amd import ops
    g, tf = model.graph, model.graph.nerf.tensorf
    n = c.shape[1]
    o, d = c.reshape(-1, 3).contiguous(), r.reshape(-1, 3).contiguous()

    def run():
        return tf.render_pose_fused(opt, o, d, var.image, ray_idx, n, white_bg=opt.nerf.setbg_opaque, ndc_ray=opt.camera.ndc,
                                    N_samples=g.nerf.n_samples)
    with torch.no_grad():
        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record()
            run()
            b.record()
        torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[10]
    rec = dict(ablate=int(os.environ.get("JT_FUSED_ABLATE", "0")), rays=int(o.shape[0]), S=int(g.nerf.n_samples),
               scene=args.scene, config=args.config, ms=round(ms, 4), loss=float(out[0]),
               shaded=None)
    print(json.dumps(rec), flush=True)
    if not args.child:
        for abl in (4, 6, 7, 15, 31):
            env = dict(os.environ, JT_FUSED_ABLATE=str(abl))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--time", "--scene", args.scene, "--config", args.config,
                            "--child"], env=env)


if __name__ == "__main__":
    main()
