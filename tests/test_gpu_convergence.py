"""Acceptance on a scene the joint optimisation can actually solve (the reference's flow: train -> Procrustes-aligned
pose error + held-out PSNR, model/bat.py:211-263, model/nerf.py:525-572).  The supervising images are RENDERED from one
known field at the ground-truth cameras (joint_tensorf_amd.synthetic.make_gt_scene / data.RenderedDataset), the run starts
from cameras perturbed by camera.noise = 0.15 (13 degrees, 0.3 scene units on average) and goes through `bat_hip.Model`'s own
lifecycle on bat_blender_VM's schedule with every iteration-denominated key divided by ten (options.compress_schedule:
all five grid stages 64^3 -> 400^3, the factor blur and the 2-D blur schedules, edge-weighted loss, optimizer rebuilds).

1. the HIP path recovers the cameras: rotation and translation error after alignment drop by more than 10 x and the
   held-out views render above 32 dB;
2. the first 200 iterations (the whole first grid stage) against the SAME loop written with the oracle's stock torch ops
   + torch.optim from the same state, host draws and jitter stream: the two pose-error curves agree."""
import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _args(**kw):
    import argparse
    d = dict(config="bat_blender_VM", compress=10.0, image_size=200, views=40, test_views=4, gt_res=128, n_voxel_final=0,
             n_rays=0, noise=None, max_iter=0, test_iter=0, seed=0, graph=False, report_every=0)
    d.update(kw)
    return argparse.Namespace(**d)


def _converge():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import converge
    return converge


def test_joint_optimisation_recovers_the_cameras():
    cv = _converge()
    opt, model = cv.build(_args(), device=DEV)
    assert opt.data.dataset_class.endswith("RenderedDataset")
    r0, t0 = cv.pose_errors(opt, model)
    assert r0 > 8.0 and t0 > 0.2  # the perturbation is real: ~13 degrees / 0.3 units
    loss = model.train(opt)
    assert model.it == 4000 and model.graph.nerf.tensorf.gridSize.tolist() == [400, 400, 400]
    r1, t1 = cv.pose_errors(opt, model)
    res = model.evaluate_full(opt)
    print("pose error after Procrustes alignment: rotation %.3f -> %.3f deg (%.0f x), translation %.4f -> %.4f (%.0f x); "
          "held-out PSNR %.2f dB %s; final loss %.2e"
          % (r0, r1, r0 / r1, t0, t1, t0 / t1, res.psnr, [round(p, 1) for p in res.psnr_per_view], float(loss.all.detach())))
    assert r1 * 10 < r0 and t1 * 10 < t0
    assert res.psnr > 32.0 and min(res.psnr_per_view) > 28.0


def test_pose_error_curve_matches_the_oracle_loop():
    """Stage 0 of the same run (200 iterations: 64^3 grid, S = 221, factor blur with the random density scale, 2-D blurred
    supervision drawn per iteration, edge-weighted loss on even iterations, L1, Adam with per-iteration lr decay, pose
    Adam + ExponentialLR) on both sides."""
    cv = _converge()
    K, EVERY = 200, 50
    opt, model = cv.build(_args(max_iter=K), device=DEV)
    g, tf = model.graph, model.graph.nerf.tensorf
    B, H, W = 40, opt.H, opt.W
    # ---- oracle side: same initial state ----
    sd = {k: v.detach().clone().contiguous() for k, v in tf.state_dict().items()}
    params = O.params_from_state_dict(sd, prefix="")
    leaves = [v for _, v in O.flat_params(params)]
    for v in leaves:
        v.requires_grad_(True)
    cfg = O.SceneCfg(opt.data.scene_bbox, tf.gridSize.tolist(), list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio).to(DEV)
    se3_o = torch.zeros(B, 6, device=DEV, requires_grad=True)
    noise = g.pose_noise.detach()
    data = model.train_data.all
    lr_i, lr_b = g.nerf.lr_index, g.nerf.lr_basis
    groups = [dict(params=params["density_line"], lr=lr_i), dict(params=params["density_plane"], lr=lr_i),
              dict(params=params["app_line"], lr=lr_i), dict(params=params["app_plane"], lr=lr_i),
              dict(params=[params["basis"]], lr=lr_b), dict(params=list(params["mlp"].values()), lr=lr_b)]
    optim_o = torch.optim.Adam(groups, betas=(0.9, 0.99))
    optim_pose_o = torch.optim.Adam([dict(params=[se3_o], lr=opt.optim.lr_pose)])
    gamma = (opt.optim.lr_pose_end / opt.optim.lr_pose) ** (1.0 / opt.max_iter)
    sched_o = torch.optim.lr_scheduler.ExponentialLR(optim_pose_o, gamma=gamma)
    decay = g.nerf.lr_decay_factor
    S = g.nerf.n_samples

    def err_o():
        with torch.no_grad():
            pose = O.train_pose(se3_o, noise, data.pose)
            al, _ = O.prealign_cameras(pose.cpu(), data.pose.cpu())
            r, t = O.camera_alignment_error(al, data.pose.cpu())
        return float(np.rad2deg(r.mean())), float(t.mean())

    # ---- HIP side ----
    curve_h = []
    orig_after = model.after_iteration

    def after(o, it=None):
        orig_after(o, it)
        if model.it % EVERY == 0:
            curve_h.append(cv.pose_errors(opt, model))
    model.after_iteration = after
    torch.manual_seed(123)
    np.random.seed(123)
    model.train(opt)
    assert model.it == K and len(curve_h) == K // EVERY

    # ---- the same loop, oracle ----
    torch.manual_seed(123)
    np.random.seed(123)
    curve_o = []
    pool2d = list(opt.c2f_alternate_2D_scale_pool)
    cache = masks = None
    for it in range(K):
        if it % 500 == 0:  # model/nerf.py:172-176 (on the host: the oracle's 2-D helpers build their taps there)
            cache = O.process_gt_images(data.image.cpu(), it / opt.max_iter, opt.blur_2d_c2f_schedule, pool2d,
                                        opt.blur_2d_c2f_kernel_size)
            masks = {k: v.to(DEV) for k, v in O.edge_masks(cache).items()}
            cache = {k: v.to(DEV) for k, v in cache.items()}
        sc = np.random.choice(pool2d)                                   # select_supervision
        image = cache[sc].view(B, 3, -1).permute(0, 2, 1)
        mask = masks[opt.edge_mask_use_scale]
        step = int(np.ceil((H * W // (int(opt.nerf.n_rays) // B)) ** 0.5))
        ox, oy = np.random.randint(step), np.random.randint(step)       # Graph.forward
        scale = np.random.choice(opt.c2f_random_density_scale_pool)     # Graph.resolve_blur
        progress = it / opt.max_iter
        optim_o.zero_grad()
        optim_pose_o.zero_grad()
        pose = O.train_pose(se3_o, noise, data.pose)
        ray_idx, _, gh, gw = O.rand_grid_ray_idx(H, W, int(opt.nerf.n_rays), B, ox, oy)
        ray_idx = ray_idx.to(DEV)
        center, ray = O.rays_for_pixels(pose, data.intr_inv, ray_idx, W)
        pd = O.interp_schedule(progress, opt.c2f_schedule_density) * scale
        pc = O.interp_schedule(progress, opt.c2f_schedule_color)
        kd = kc = None
        if max(pd, pc) >= 0.001:
            kd, kc = O.get_kernel(cfg, pd, opt.c2f_kernel_size).to(DEV), O.get_kernel(cfg, pc, opt.c2f_kernel_size).to(DEV)
        jit = torch.rand(B * gh * gw, 1, device=DEV)                     # the draw BAT_VMSplit.forward takes
        rgb, _, _ = O.render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), S, white_bg=True, jitter=jit,
                             kernel_density=kd, kernel_color=kc)
        rgb = rgb.view(B, -1, 3)
        if it % 2 == 0 and it < opt.edge_mask_before_iter:
            render = O.render_loss(rgb, image[:, ray_idx], mask[:, ray_idx], opt.edge_loss_factor, opt.non_edge_loss_factor)
        else:
            render = O.render_loss(rgb, image[:, ray_idx])
        total = float(opt.loss_weight.render) * render + float(opt.loss_weight.L1.init) * O.density_L1(params)
        total.backward()
        optim_o.step()
        optim_pose_o.step()
        sched_o.step()
        for grp in optim_o.param_groups:
            grp["lr"] *= decay
        if (it + 1) % EVERY == 0:
            curve_o.append(err_o())
    print("rotation error [deg]    hip   ", [round(r, 3) for r, _ in curve_h])
    print("                        oracle", [round(r, 3) for r, _ in curve_o])
    print("translation error       hip   ", [round(t, 4) for _, t in curve_h])
    print("                        oracle", [round(t, 4) for _, t in curve_o])
    # same state, same draws: the two runs separate only through round-off amplified by Adam (tests/test_gpu_trajectory.py
    # bounds that over 8 iterations); over 200 iterations the pose-error curves stay together
    for (rh, th), (ro, to) in zip(curve_h, curve_o):
        assert abs(rh - ro) <= 0.1 * ro + 0.05, (curve_h, curve_o)
        assert abs(th - to) <= 0.1 * to + 0.002, (curve_h, curve_o)
    assert curve_o[-1][0] < 0.8 * curve_o[0][0] or curve_o[-1][0] < 12.0  # and the oracle's own run is converging too


# ---------------------------------------------------------------------------------------------------------------------
# bat_llff_VM_MLP (BASELINE.json configs[2]): the forward-facing configuration
# ---------------------------------------------------------------------------------------------------------------------
# The forward-facing synthetic scene (synthetic.make_gt_scene, NDC): forty cameras on a 0.3-wide patch looking along +z, six
# textured blobs (radius 0.15 - 0.35) at depth 1.2 - 2 near-plane distances, and behind them a textured back wall in EIGHT DEPTH
# STEPS across the picture (depth 1.8 - 19; synthetic.bake_stairs).  Round 3's scene closed the picture with ONE fronto-parallel
# wall: content at a single depth, for which a sideways translation and a small rotation of a camera are the same image motion --
# every variant of it ended with cameras rotated by 35 - 160 degrees and held-out views at 9 - 17 dB
# (profiles/round4_llff_convergence.txt); with depth steps the same run recovers the camera centres and renders held-out views.
LLFF_SCENE = dict(config="bat_llff_VM_MLP", views=40, image_size=240, llff_baseline=0.3, llff_focus=0.0, gt_z_range="0.35,0.6",
                  gt_stairs=8, gt_blobs=6, gt_blob_radius="0.15,0.35")


def test_llff_joint_optimisation_recovers_camera_centres():
    """bat_llff_VM_MLP's whole schedule (compressed ten times: 5 000 iterations, all five grid stages up to 771 x 859 x 771,
    hipGraph replay) on the forward-facing scene, from IDENTITY poses (the reference's initialisation, model/bat.py:341-353):
    the camera centres after Procrustes alignment come out closer to the truth than the start (all cameras at the origin) by
    more than 1.5 x, and the held-out views render above 23 dB without test-time optimisation.  Measured over three runs
    (the order of the float atomics differs from run to run and 5 000 Adam steps amplify it): 1.8 - 2.7 x (3.1 x once),
    25.6 - 29.5 dB; the rotations, exact at the start because the true cameras do not rotate, drift by 4 - 10 degrees while the
    factors are blurred -- the residual of the translation / rotation coupling of a forward-facing capture.  This is NOT the
    >= 5 x / >= 25 dB the round-3 verdict asked for; what is established instead is attribution: the pose-error curve of the HIP
    path equals the oracle loop's (next test), so whatever the forward-facing run does, the reference algorithm does too."""
    cv = _converge()
    opt, model = cv.build(_args(graph=True, **LLFF_SCENE), device=DEV)
    r0, t0 = cv.pose_errors(opt, model)
    loss = model.train(opt)
    assert model.it == 5000 and model.graph.nerf.tensorf.gridSize.tolist() == [771, 859, 771]
    r1, t1 = cv.pose_errors(opt, model)
    res = model.evaluate_full(opt)
    print("LLFF pose error after Procrustes alignment: rotation %.3f -> %.3f deg, translation %.4f -> %.4f (%.1f x); held-out "
          "PSNR %.2f dB %s; final loss %.2e" % (r0, r1, t0, t1, t0 / t1, res.psnr, [round(p, 1) for p in res.psnr_per_view],
                                                float(loss.all.detach())))
    assert t1 * 1.5 < t0 and r1 < 15.0
    assert res.psnr > 23.0


def _llff_oracle_loop(cv, opt, model, K, EVERY):
    """K iterations of bat_llff_VM_MLP's training loop (model/nerf.py:150-278 around model/bat.py:96-116) written with the
    oracle's stock torch ops + torch.optim on the GPU, from `model`'s CURRENT state and the process's CURRENT random
    streams: NDC rays from identity-initialised poses, the shared jittered z row, relu density, WeakView MLP, the near-plane
    schedule, the white-background coin, factor blur with the random density scale, 2-D blurred supervision, edge-weighted
    loss on even iterations, L1 + TV with decaying weights, Adam with per-iteration lr decay, pose Adam stepping every
    n_AccumPoseGrad-th iteration on accumulated gradients with warm-up and ExponentialLR, and the pose reset of
    train_schedule.reset_pose_on_iter.  Returns the pose-error curve [(rotation deg, translation)] every EVERY iterations."""
    g, tf = model.graph, model.graph.nerf.tensorf
    data = model.train_data.all
    B, H, W = len(data.idx), opt.H, opt.W
    sd = {k: v.detach().clone().contiguous() for k, v in tf.state_dict().items()}
    params = O.params_from_state_dict(sd, prefix="")
    for _, v in O.flat_params(params):
        v.requires_grad_(True)
    cfg = O.SceneCfg(opt.data.scene_bbox, tf.gridSize.tolist(), list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio,
                     density_shift=float(opt.arch.density_shift), distance_scale=float(opt.arch.distance_scale),
                     fea2denseAct="relu", rayMarch_weight_thres=float(opt.arch.tensorf.rayMarch_weight_thres),
                     shadingMode="MLP_Fea_WeakView", view_pe=2, fea_pe=2, ndc_near_plane=float(opt.arch.ndc_near_plane)).to(DEV)
    se3_o = g.se3_refine.weight.detach().clone().requires_grad_(True)
    eye = torch.eye(3, 4, device=DEV)
    lr_i, lr_b = g.nerf.lr_index, g.nerf.lr_basis
    groups = [dict(params=params["density_line"], lr=lr_i), dict(params=params["density_plane"], lr=lr_i),
              dict(params=params["app_line"], lr=lr_i), dict(params=params["app_plane"], lr=lr_i),
              dict(params=[params["basis"]], lr=lr_b), dict(params=list(params["mlp"].values()), lr=lr_b)]
    optim_o = torch.optim.Adam(groups, betas=(0.9, 0.99))
    optim_pose_o = torch.optim.Adam([dict(params=[se3_o], lr=opt.optim.lr_pose)])
    gamma = (opt.optim.lr_pose_end / opt.optim.lr_pose) ** (1.0 / opt.max_iter)
    sched_o = torch.optim.lr_scheduler.ExponentialLR(optim_pose_o, gamma=gamma)
    decay = g.nerf.lr_decay_factor
    S = g.nerf.n_samples
    ts = opt.train_schedule
    w_tvd, w_tvc = float(opt.loss_weight.TV_density), float(opt.loss_weight.TV_color)
    pool2d = list(opt.c2f_alternate_2D_scale_pool)

    def err_o():
        with torch.no_grad():
            pose = O.train_pose(se3_o, None, eye)
            al, _ = O.prealign_cameras(pose.cpu(), data.pose.cpu())
            r, t = O.camera_alignment_error(al, data.pose.cpu())
        return float(np.rad2deg(r.mean())), float(t.mean())

    curve, cache, masks = [], None, None
    for it in range(K):
        n_rays = int(ts.n_rays_init if it < ts.change_n_rays_after_n_iters else ts.n_rays_rest)          # before_iteration
        accum = int(ts.n_AccumPoseGrad_init if it < ts.change_n_AccumPoseGrad_after_n_iters else ts.n_AccumPoseGrad_rest)
        if it == ts.reset_pose_on_iter:
            with torch.no_grad():
                se3_o.mul_(0.0)
        if it % 500 == 0:                                                                               # select_supervision
            cache = O.process_gt_images(data.image.cpu(), it / opt.max_iter, opt.blur_2d_c2f_schedule, pool2d,
                                        opt.blur_2d_c2f_kernel_size)
            masks = {k: v.to(DEV) for k, v in O.edge_masks(cache).items()}
            cache = {k: v.to(DEV) for k, v in cache.items()}
        sc = np.random.choice(pool2d)
        image = cache[sc].view(B, 3, -1).permute(0, 2, 1)
        mask = masks[opt.edge_mask_use_scale]
        for pg in optim_pose_o.param_groups:                                                            # train_iteration
            pg["lr_orig"] = pg["lr"]
            pg["lr"] *= min(1, it / opt.optim.warmup_pose)
        optim_o.zero_grad()
        pose = O.train_pose(se3_o, None, eye)
        step = int(np.ceil((H * W // (n_rays // B)) ** 0.5))
        ox, oy = np.random.randint(step), np.random.randint(step)
        ray_idx, _, gh, gw = O.rand_grid_ray_idx(H, W, n_rays, B, ox, oy)
        ray_idx = ray_idx.to(DEV)
        center, ray = O.rays_for_pixels(pose, data.intr_inv, ray_idx, W)
        center, ray = O.convert_ndc(center, ray, data.intr, near=float(opt.arch.ndc_near_plane))
        scale = np.random.choice(opt.c2f_random_density_scale_pool)
        progress = it / opt.max_iter
        cfg.near_far[0] = O.interp_schedule(progress, opt.tensorf_near_plane_schedule)
        pd = O.interp_schedule(progress, opt.c2f_schedule_density) * scale
        pc = O.interp_schedule(progress, opt.c2f_schedule_color)
        kd = kc = None
        if max(pd, pc) >= 0.001:
            kd, kc = O.get_kernel(cfg, pd, opt.c2f_kernel_size).to(DEV), O.get_kernel(cfg, pc, opt.c2f_kernel_size).to(DEV)
        jit = torch.rand(1, S, device=DEV)                      # the draws BAT_VMSplit.forward takes: z row, then the coin
        coin = float(torch.rand((1,)))
        rgb, _, _ = O.render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), S, white_bg=coin < 0.5, jitter=jit,
                             ndc_ray=True, kernel_density=kd, kernel_color=kc)
        rgb = rgb.view(B, -1, 3)
        if it % 2 == 0 and it < opt.edge_mask_before_iter:
            render = O.render_loss(rgb, image[:, ray_idx], mask[:, ray_idx], opt.edge_loss_factor, opt.non_edge_loss_factor)
        else:
            render = O.render_loss(rgb, image[:, ray_idx])
        w_l1 = float(opt.loss_weight.L1.rest if it > ts.update_alphamask_iters[0] else opt.loss_weight.L1.init)
        total = float(opt.loss_weight.render) * render + w_l1 * O.density_L1(params) \
            + w_tvd * O.tv_planes(params["density_plane"]) + w_tvc * O.tv_planes(params["app_plane"])
        total.backward()
        optim_o.step()
        if (it + 1) % accum == 0:                               # pose gradients accumulate over `accum` iterations
            optim_pose_o.step()
            optim_pose_o.zero_grad()
        for pg in optim_pose_o.param_groups:
            pg["lr"] = pg["lr_orig"]
        sched_o.step()
        for grp in optim_o.param_groups:
            grp["lr"] *= decay
        w_tvd *= decay
        w_tvc *= decay
        if (it + 1) % EVERY == 0:
            curve.append(err_o())
    return curve


def _hip_curve(cv, opt, model, K, EVERY):
    curve = []
    orig_after = model.after_iteration

    def after(o, it=None):
        orig_after(o, it)
        if model.it % EVERY == 0:
            curve.append(cv.pose_errors(opt, model))
    model.after_iteration = after
    model.train(opt)
    model.after_iteration = orig_after
    assert model.it == K and len(curve) == K // EVERY
    return curve


def test_llff_pose_error_curve_matches_the_oracle_loop():
    """bat_llff_VM_MLP's joint optimisation through the HIP path against the SAME loop written with the
    oracle, 300 iterations of the schedule compressed ten times (the whole first grid stage: 20 480-nominal-ray lattices,
    factor blur, the near-plane schedule, pose steps every 8th iteration, warm-up, the pose reset at iteration 250) from the
    same state and the same host / device draws: whatever the forward-facing run does to its cameras, the reference
    algorithm in stock torch ops does the same."""
    cv = _converge()
    K, EVERY = 300, 50
    opt, model = cv.build(_args(max_iter=K, **LLFF_SCENE), device=DEV)
    assert bool(opt.camera.ndc) and opt.train_schedule.reset_pose_on_iter == 250 and opt.optim.warmup_pose == 50
    torch.manual_seed(321)
    np.random.seed(321)
    rng_state = (torch.get_rng_state(), torch.cuda.get_rng_state(), np.random.get_state())
    # the oracle loop first (it starts from the model's initial state and does not touch it), then the HIP run on the same streams
    curve_o = _llff_oracle_loop(cv, opt, model, K, EVERY)
    torch.set_rng_state(rng_state[0])
    torch.cuda.set_rng_state(rng_state[1])
    np.random.set_state(rng_state[2])
    curve_h = _hip_curve(cv, opt, model, K, EVERY)
    print("rotation error [deg]    hip   ", [round(r, 3) for r, _ in curve_h])
    print("                        oracle", [round(r, 3) for r, _ in curve_o])
    print("translation error       hip   ", [round(t, 4) for _, t in curve_h])
    print("                        oracle", [round(t, 4) for _, t in curve_o])
    for (rh, th), (ro, to) in zip(curve_h, curve_o):
        assert abs(rh - ro) <= 0.15 * ro + 0.05, (curve_h, curve_o)
        assert abs(th - to) <= 0.15 * to + 0.003, (curve_h, curve_o)


def _oracle_rendered(model, opt):
    """Replace the scene's forward (BAT_VMSplit.forward -> ops.render_rays: blur, march, shade, compositing and their autograd
    through the HIP kernels) by the oracle's stock-torch render of the SAME live Parameters, taking the same random draws in
    the same order (shared z jitter row, then the white-background coin).  Everything around it -- schedule, lattice, ray
    generation, losses, regularisers, Adam, upsampling, alpha mask -- stays bat_hip.Model's."""
    tf = model.graph.nerf.tensorf

    def forward(opt_, center, ray_dir, white_bg=True, is_train=False, ndc_ray=False, N_samples=-1,
                c2f_parameter_density=None, c2f_parameter_color=None, c2f_mode=None, c2f_kernel_size=None,
                is_test_optim=False, view_pe_progress=1.0, fea_pe_progress=1.0):
        tf.__dict__.setdefault("_reg_cache", {}).clear()
        S = N_samples if N_samples > 0 else tf.nSamples
        cfg = O.SceneCfg(opt.data.scene_bbox, tf.gridSize.tolist(), [float(tf.near_far[0]), float(tf.near_far[1])],
                         step_ratio=opt.nerf.step_ratio, density_shift=float(opt.arch.density_shift),
                         distance_scale=float(opt.arch.distance_scale), fea2denseAct="relu",
                         rayMarch_weight_thres=float(opt.arch.tensorf.rayMarch_weight_thres), shadingMode="MLP_Fea_WeakView",
                         view_pe=2, fea_pe=2, ndc_near_plane=float(opt.arch.ndc_near_plane)).to(DEV)
        params = O.params_from_state_dict(dict(tf.named_parameters()), prefix="")
        jit = torch.rand(1, S, device=DEV) if (is_train and ndc_ray) else None
        kd = kc = None
        if c2f_mode is not None:
            kd = O.get_kernel(cfg, c2f_parameter_density, c2f_kernel_size).to(DEV)
            kc = O.get_kernel(cfg, c2f_parameter_color, c2f_kernel_size).to(DEV)
        wb = True if white_bg else (float(torch.rand((1,))) < 0.5 if is_train else False)
        am = None
        if tf.alphaMask is not None:
            am = (tf.alphaMask.alpha_volume, tf.alphaMask.aabb)
        rgb, depth, opacity = O.render(cfg, params, center, ray_dir, S, white_bg=wb, jitter=jit, ndc_ray=ndc_ray,
                                       kernel_density=kd, kernel_color=kc, alpha_mask=am)
        return rgb, depth, opacity

    tf.forward = forward
    return tf


def _hip_and_oracle_rendered_runs(scene_args, every, no_alpha_mask=False):
    """The same compressed bat_llff_VM_MLP run twice from the same seed: through the HIP renderer, and with `BAT_VMSplit.forward`
    replaced by the oracle's stock torch ops + torch autograd (`_oracle_rendered`).  Returns {"hip": ..., "oracle": ...} with
    the start / end pose errors, the pose-error curve every `every` iterations, the held-out PSNR (evaluated through the
    product path on both sides) and the final grid."""
    import json
    cv = _converge()
    out = {}
    for name in ("hip", "oracle"):
        opt, model = cv.build(_args(graph=False, **scene_args), device=DEV)
        if no_alpha_mask:
            # (a final grid below 256^3 would trigger the alpha-mask update + AABB shrink of model/tensorf.py:480-489, which the
            #  BAT yamls' own schedules never reach and the stock-op stand-in does not carry: keep the first entry -- the L1
            #  weight switches there -- and move the updates themselves past the end of the run)
            first = opt.train_schedule.update_alphamask_iters[0]
            opt.train_schedule.update_alphamask_iters = [first] + [10 ** 9]
            model.graph.nerf._update_alphamask = lambda *a, **k: None
        r0, t0 = cv.pose_errors(opt, model)
        tf = model.graph.nerf.tensorf
        if name == "oracle":
            _oracle_rendered(model, opt)
        curve = []
        orig_after = model.after_iteration

        def after(o, it=None, model=model, opt=opt, curve=curve, orig_after=orig_after):
            orig_after(o, it)
            if model.it % every == 0:
                curve.append((model.it,) + tuple(round(v, 4) for v in cv.pose_errors(opt, model)))
        model.after_iteration = after
        model.train(opt)
        r1, t1 = cv.pose_errors(opt, model)
        rrel = cv.relative_rotation_error(opt, model)
        if name == "oracle":
            del tf.forward                      # evaluation of the trained state through the product path on both sides
        res = model.evaluate_full(opt)
        out[name] = dict(rot_deg=(round(r0, 3), round(r1, 3)), rot_rel_deg_end=round(rrel, 3), trans=(round(t0, 4), round(t1, 4)),
                         trans_gain=round(t0 / t1, 2), psnr=round(res.psnr, 2), grid=tf.gridSize.tolist(), curve=curve,
                         iterations=model.it)
        print(name, json.dumps(out[name]), flush=True)
        del model
        torch.cuda.empty_cache()
    return out


@pytest.mark.skipif(__import__("os").environ.get("JT_LONG_TESTS") != "1", reason="twenty minutes: set JT_LONG_TESTS=1")
def test_llff_full_schedule_with_the_oracle_render_ends_where_the_hip_path_ends():
    """Round-3 verdict item 3, the alternative it allows: "or commit the oracle-side curve showing the reference algorithm fails
    identically".  The WHOLE compressed bat_llff_VM_MLP schedule (5 000 iterations, five grid stages) on LLFF_SCENE twice from
    the same seed: once through the HIP renderer, once with the renderer replaced by the oracle's stock torch ops + torch
    autograd (`_oracle_rendered`).  Both runs are chaotic in their details (Adam amplifies round-off over 5 000 steps), so the
    assertion is on where they END: camera-centre recovery within a factor two of each other, held-out PSNR within 6 dB,
    neither reaching the 5 x the verdict asked of the scene."""
    out = _hip_and_oracle_rendered_runs(LLFF_SCENE, 500)
    h, o = out["hip"], out["oracle"]
    assert h["grid"] == o["grid"] == [771, 859, 771]
    assert 0.5 < h["trans_gain"] / o["trans_gain"] < 2.0
    assert abs(h["psnr"] - o["psnr"]) < 6.0
    assert o["trans_gain"] < 5.0


def test_llff_short_schedule_with_the_oracle_render_ends_where_the_hip_path_ends():
    """The same comparison in a form the default suite can afford: the schedule compressed fifty times
    (1 000 iterations through all five grid stages, 2 048 nominal rays, the final grid capped at 2 M voxels so that the stock-op
    renderer stays in seconds; the alpha-mask updates such a small grid would trigger are switched off on both sides), 20 views
    of 120 pixels.  Measured at compress 40 / schedule rays: camera-centre error 0.1787 (HIP) / 0.1773 (oracle render), every
    point of the two curves within 2 %, relative rotations 1.7 / 1.4 degrees, 18.7 / 19.8 dB.  At this length neither run recovers much; what is asserted is that the two renderers
    take the joint optimisation to the same place: camera-centre error within 25 % of each other at the end and at every
    recorded point of the curve, relative rotations within three degrees (held-out PSNR: see the note at the assertion)."""
    scene = dict(LLFF_SCENE, views=20, image_size=120, compress=50.0, n_voxel_final=2000000, n_rays=2048)
    out = _hip_and_oracle_rendered_runs(scene, 250, no_alpha_mask=True)
    h, o = out["hip"], out["oracle"]
    assert h["iterations"] == o["iterations"] == 1000 and h["grid"] == o["grid"]
    assert abs(h["trans"][1] - o["trans"][1]) <= 0.25 * o["trans"][1], (h, o)
    for (ih, rh, th), (io, ro, to) in zip(h["curve"], o["curve"]):
        assert ih == io and abs(th - to) <= 0.25 * to + 0.002, (h["curve"], o["curve"])
    # (the two runs are two samples of a chaotic optimisation -- float atomics on both sides -- that agree to 0.1 % at iteration
    #  250 and drift apart afterwards; relative rotations at the end, HIP / oracle render, on five boxes: 1.7 / 1.4, 1.77 / 1.86,
    #  1.77 / 1.71, 1.77 / 1.71 and 3.59 / 1.80 degrees)
    assert h["rot_rel_deg_end"] < 6.0 and o["rot_rel_deg_end"] < 6.0 and abs(h["rot_rel_deg_end"] - o["rot_rel_deg_end"]) <= 3.0, (h, o)
    # Held-out PSNR of a run this short is NOT compared: with the cameras still 0.16 off, what a held-out view renders at is
    # decided by where its own test-time pose search happens to land -- HIP / oracle-render pairs seen on four boxes: 18.7 / 19.8,
    # 11.1 / 9.9, 14.2 / 10.8 and 10.6 / 22.9 dB, i.e. 8-23 dB on EITHER side.  Both must be finite renders of the scene, nothing
    # more; the long form of this test (JT_LONG_TESTS), where the schedule is complete, holds the two within 6 dB.
    assert 5.0 < h["psnr"] < 60.0 and 5.0 < o["psnr"] < 60.0, (h, o)
