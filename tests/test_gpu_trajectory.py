"""End-to-end trajectory parity: K full training iterations (pose composition, lattice, blurred factors with the
random scale, stratified jitter, edge-weighted loss on alternate iterations, L1, Adam on six parameter groups with
the per-iteration lr decay, pose Adam + ExponentialLR, progress / schedule update) through bat_hip.Model on the
GPU against the same loop written with the CPU oracle + torch.optim, from the same initial state and with the same
host / device random draws."""
import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
K = 8


def _oracle_params(model):
    sd = {k: v.detach().cpu().clone() for k, v in model.graph.nerf.tensorf.state_dict().items()}
    p = O.params_from_state_dict(sd, prefix="")
    for _, v in O.flat_params(p):
        v.requires_grad_(True)
    return p


def test_training_trajectory_matches_oracle_loop():
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views
    B, HW = 3, 40
    opt = make_options("bat_blender_VM", device=DEV, data=dict(image_size=[HW, HW], num_views=B),
                       train_schedule=dict(n_voxel_init=14 ** 3, n_rays_init=96, n_rays_rest=96), nerf=dict(n_rays=96))
    torch.manual_seed(0)
    np.random.seed(0)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=B)
    model.setup_optimizer(opt)
    g = model.graph
    tf = g.nerf.tensorf
    with torch.no_grad():  # semi-transparent content instead of the near-empty initial field
        for p in tf.density_plane:
            p.mul_(22.0)
        g.se3_refine.weight.copy_(0.01 * torch.randn(B, 6, device=DEV))
    var0 = make_views(opt, B, seed=3, device=DEV)

    # ---------------- oracle side: same state, its own optimisers ----------------
    params = _oracle_params(model)
    cfg = O.SceneCfg(opt.data.scene_bbox, tf.gridSize.tolist(), list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio)
    se3_o = g.se3_refine.weight.detach().cpu().clone().requires_grad_(True)
    noise = g.pose_noise.detach().cpu()
    lr_i, lr_b = g.nerf.lr_index, g.nerf.lr_basis
    groups = [dict(params=params["density_line"], lr=lr_i), dict(params=params["density_plane"], lr=lr_i),
              dict(params=params["app_line"], lr=lr_i), dict(params=params["app_plane"], lr=lr_i),
              dict(params=[params["basis"]], lr=lr_b), dict(params=list(params["mlp"].values()), lr=lr_b)]
    optim_o = torch.optim.Adam(groups, betas=(0.9, 0.99))
    optim_pose_o = torch.optim.Adam([dict(params=[se3_o], lr=opt.optim.lr_pose)])
    gamma = (opt.optim.lr_pose_end / opt.optim.lr_pose) ** (1.0 / opt.max_iter)
    sched_o = torch.optim.lr_scheduler.ExponentialLR(optim_pose_o, gamma=gamma)
    decay = g.nerf.lr_decay_factor
    cpu = {k: v.cpu() for k, v in dict(var0).items() if torch.is_tensor(v)}
    image = cpu["image"].view(B, 3, -1).permute(0, 2, 1)

    # ---------------- the draws both sides consume ----------------
    rs = np.random.RandomState(7)
    gj = torch.Generator().manual_seed(11)
    offs = [(int(rs.randint(5)), int(rs.randint(5))) for _ in range(K)]
    pool = list(opt.c2f_random_density_scale_pool)
    scales = [float(pool[rs.randint(len(pool))]) for _ in range(K)]

    loss_hip, loss_ora = [], []
    orig_randint, orig_choice = np.random.randint, np.random.choice
    try:
        for it in range(K):
            # ---- HIP: one Model.train_iteration with the draws injected ----
            ints, ch = list(offs[it]), [scales[it]]
            np.random.randint = lambda *a, **k: ints.pop(0)
            np.random.choice = lambda *a, **k: ch.pop(0)
            step = int(np.ceil((HW * HW // (96 // B)) ** 0.5))
            assert offs[it][0] < step and offs[it][1] < step
            n_lattice = len(range(offs[it][0], HW, step)) * len(range(offs[it][1], HW, step))
            jit = torch.rand(B * n_lattice, 1, generator=gj)
            tf.jitter_override = jit.to(DEV)
            loss = model.train_iteration(opt, Opt(dict(var0)))
            model.after_iteration(opt)
            loss_hip.append(float(loss.all.detach()))
            assert not ints and not ch

            # ---- oracle: the same iteration ----
            optim_o.zero_grad()
            optim_pose_o.zero_grad()
            progress = it / opt.max_iter
            pose = O.train_pose(se3_o, noise, cpu["pose"])
            ray_idx, _, gh, gw = O.rand_grid_ray_idx(HW, HW, 96, B, offs[it][0], offs[it][1])
            center, ray = O.rays_for_pixels(pose, cpu["intr_inv"], ray_idx, HW)
            pd = O.interp_schedule(progress, opt.c2f_schedule_density) * scales[it]
            pc = O.interp_schedule(progress, opt.c2f_schedule_color)
            kd = kc = None
            if max(pd, pc) >= 0.001:
                kd, kc = O.get_kernel(cfg, pd, opt.c2f_kernel_size), O.get_kernel(cfg, pc, opt.c2f_kernel_size)
            rgb, depth, acc = O.render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), g.nerf.n_samples,
                                       white_bg=True, jitter=jit, kernel_density=kd, kernel_color=kc)
            rgb = rgb.view(B, -1, 3)
            if it % 2 == 0 and it < opt.edge_mask_before_iter:
                render = O.render_loss(rgb, image[:, ray_idx], cpu["train_edge_masks"][:, ray_idx], opt.edge_loss_factor,
                                       opt.non_edge_loss_factor)
            else:
                render = O.render_loss(rgb, image[:, ray_idx])
            total = float(opt.loss_weight.render) * render + float(opt.loss_weight.L1.init) * O.density_L1(params)
            total.backward()
            optim_o.step()
            optim_pose_o.step()
            sched_o.step()
            for grp in optim_o.param_groups:
                grp["lr"] *= decay
            loss_ora.append(float(total.detach()))
    finally:
        np.random.randint, np.random.choice = orig_randint, orig_choice
        tf.jitter_override = None
    print("loss (hip)   ", np.round(loss_hip, 6))
    print("loss (oracle)", np.round(loss_ora, 6))
    # iteration 0 is a pure forward comparison; later ones include everything the optimisers did in between.
    # Adam turns round-off in a near-zero gradient into a full +-lr step of that element, so the trajectories
    # separate slowly: 2e-4 relative over 8 iterations (measured: identical to 6 digits).
    np.testing.assert_allclose(loss_hip[0], loss_ora[0], rtol=2e-5)
    np.testing.assert_allclose(loss_hip, loss_ora, rtol=2e-4)
    # parameters after K steps
    sd = {k: v.detach().cpu() for k, v in tf.state_dict().items()}
    ph = O.params_from_state_dict(sd, prefix="")
    for (n, a), (_, b) in zip(O.flat_params(ph), O.flat_params(params)):
        diff = float((a - b.detach()).abs().max())
        # an element whose gradient is round-off can move by +-lr per step in opposite directions
        assert diff <= 2.2 * K * max(lr_i, lr_b), (n, diff)
        assert float((a - b.detach()).abs().mean()) <= 2e-4, (n, float((a - b.detach()).abs().mean()))
    d = (g.se3_refine.weight.detach().cpu() - se3_o.detach()).abs().max()
    assert float(d) <= 2e-4, float(d)


def test_training_trajectory_llff_matches_oracle_loop():
    """The same for bat_llff_VM_MLP: NDC rays with the shared jittered z row, relu density, WeakView MLP, non-cubic
    grid (blur through the reference's reshape quirk), the near-plane schedule, the CPU coin of the white
    background, TV regularisers whose weights decay every iteration, pose-lr warm-up."""
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views
    B, H, W, NR = 3, 30, 40, 90
    opt = make_options("bat_llff_VM_MLP", device=DEV, data=dict(image_size=[H, W], num_views=B),
                       train_schedule=dict(n_voxel_init=2200, n_rays_init=NR, n_rays_rest=NR), nerf=dict(n_rays=NR))
    torch.manual_seed(1)
    np.random.seed(1)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=B)
    model.setup_optimizer(opt)
    model.it = 40                       # inside the pose-lr warm-up (it / 500), edge loss on alternate iterations
    g = model.graph
    g.nerf.set_progress(model.it / opt.max_iter)
    tf = g.nerf.tensorf
    with torch.no_grad():
        g.se3_refine.weight.copy_(0.01 * torch.randn(B, 6, device=DEV))
    var0 = make_views(opt, B, seed=4, device=DEV)

    params = _oracle_params(model)
    grid = tf.gridSize.tolist()
    assert len(set(grid)) > 1  # non-cubic
    cfg = O.SceneCfg(opt.data.scene_bbox, grid, list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio,
                     density_shift=float(opt.arch.density_shift), distance_scale=float(opt.arch.distance_scale),
                     fea2denseAct="relu", rayMarch_weight_thres=float(opt.arch.tensorf.rayMarch_weight_thres),
                     shadingMode="MLP_Fea_WeakView", view_pe=2, fea_pe=2, ndc_near_plane=float(opt.arch.ndc_near_plane))
    se3_o = g.se3_refine.weight.detach().cpu().clone().requires_grad_(True)
    lr_i, lr_b = g.nerf.lr_index, g.nerf.lr_basis
    groups = [dict(params=params["density_line"], lr=lr_i), dict(params=params["density_plane"], lr=lr_i),
              dict(params=params["app_line"], lr=lr_i), dict(params=params["app_plane"], lr=lr_i),
              dict(params=[params["basis"]], lr=lr_b), dict(params=list(params["mlp"].values()), lr=lr_b)]
    optim_o = torch.optim.Adam(groups, betas=(0.9, 0.99))
    optim_pose_o = torch.optim.Adam([dict(params=[se3_o], lr=opt.optim.lr_pose)])
    gamma = (opt.optim.lr_pose_end / opt.optim.lr_pose) ** (1.0 / opt.max_iter)
    sched_o = torch.optim.lr_scheduler.ExponentialLR(optim_pose_o, gamma=gamma)
    decay = g.nerf.lr_decay_factor
    cpu = {k: v.cpu() for k, v in dict(var0).items() if torch.is_tensor(v)}
    image = cpu["image"].view(B, 3, -1).permute(0, 2, 1)
    w_tvd, w_tvc = float(opt.loss_weight.TV_density), float(opt.loss_weight.TV_color)
    S = g.nerf.n_samples
    eye = torch.eye(3, 4)

    rs = np.random.RandomState(9)
    gj = torch.Generator().manual_seed(13)
    step = int(np.ceil((H * W // (NR // B)) ** 0.5))
    offs = [(int(rs.randint(step)), int(rs.randint(step))) for _ in range(K)]
    pool = list(opt.c2f_random_density_scale_pool)
    scales = [float(pool[1 + rs.randint(len(pool) - 1)]) for _ in range(K)]
    coins = [float(rs.rand()) for _ in range(K)]

    loss_hip, loss_ora = [], []
    orig_randint, orig_choice = np.random.randint, np.random.choice
    it0 = model.it
    try:
        for k in range(K):
            it = it0 + k
            ints, ch = list(offs[k]), [scales[k]]
            np.random.randint = lambda *a, **kw: ints.pop(0)
            np.random.choice = lambda *a, **kw: ch.pop(0)
            jit = torch.rand(1, S, generator=gj)
            tf.jitter_override = jit.to(DEV)
            tf.coin_override = coins[k]
            loss = model.train_iteration(opt, Opt(dict(var0)))
            model.after_iteration(opt)
            loss_hip.append(float(loss.all.detach()))
            assert not ints and not ch

            optim_o.zero_grad()
            optim_pose_o.zero_grad()
            progress = it / opt.max_iter
            for pg in optim_pose_o.param_groups:  # warm-up (model/bat.py:99-103): lr * min(1, it / warmup) for this step
                pg["lr_orig"] = pg["lr"]
                pg["lr"] *= min(1, it / opt.optim.warmup_pose)
            pose = O.train_pose(se3_o, None, eye)
            ray_idx, _, gh, gw = O.rand_grid_ray_idx(H, W, NR, B, offs[k][0], offs[k][1])
            center, ray = O.rays_for_pixels(pose, cpu["intr_inv"], ray_idx, W)
            center, ray = O.convert_ndc(center, ray, cpu["intr"], near=float(opt.arch.ndc_near_plane))
            cfg.near_far[0] = O.interp_schedule(progress, opt.tensorf_near_plane_schedule)
            pd = O.interp_schedule(progress, opt.c2f_schedule_density) * scales[k]
            pc = O.interp_schedule(progress, opt.c2f_schedule_color)
            kd = kc = None
            if max(pd, pc) >= 0.001:
                kd, kc = O.get_kernel(cfg, pd, opt.c2f_kernel_size), O.get_kernel(cfg, pc, opt.c2f_kernel_size)
            rgb, depth, acc = O.render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), S,
                                       white_bg=coins[k] < 0.5, jitter=jit, ndc_ray=True, kernel_density=kd,
                                       kernel_color=kc)
            rgb = rgb.view(B, -1, 3)
            if it % 2 == 0 and it < opt.edge_mask_before_iter:
                render = O.render_loss(rgb, image[:, ray_idx], cpu["train_edge_masks"][:, ray_idx], opt.edge_loss_factor,
                                       opt.non_edge_loss_factor)
            else:
                render = O.render_loss(rgb, image[:, ray_idx])
            total = float(opt.loss_weight.render) * render + float(opt.loss_weight.L1.init) * O.density_L1(params) \
                + w_tvd * O.tv_planes(params["density_plane"]) + w_tvc * O.tv_planes(params["app_plane"])
            total.backward()
            optim_o.step()
            optim_pose_o.step()
            for pg in optim_pose_o.param_groups:
                pg["lr"] = pg["lr_orig"]
            sched_o.step()
            for grp in optim_o.param_groups:
                grp["lr"] *= decay
            w_tvd *= decay
            w_tvc *= decay
            loss_ora.append(float(total.detach()))
    finally:
        np.random.randint, np.random.choice = orig_randint, orig_choice
        tf.jitter_override = None
        tf.coin_override = None
    print("loss (hip)   ", np.round(loss_hip, 6))
    print("loss (oracle)", np.round(loss_ora, 6))
    np.testing.assert_allclose(loss_hip[0], loss_ora[0], rtol=2e-5)
    np.testing.assert_allclose(loss_hip, loss_ora, rtol=3e-4)
    sd = {k2: v.detach().cpu() for k2, v in tf.state_dict().items()}
    ph = O.params_from_state_dict(sd, prefix="")
    for (n, a), (_, b) in zip(O.flat_params(ph), O.flat_params(params)):
        assert float((a - b.detach()).abs().mean()) <= 3e-4, (n, float((a - b.detach()).abs().mean()))
    d = (g.se3_refine.weight.detach().cpu() - se3_o.detach()).abs().max()
    assert float(d) <= 2e-4, float(d)


def test_plain_tensorf_trains_through_an_alpha_mask_update():
    """`model=tensorf` (known poses, TensorVMSplit): a few iterations across an alpha-mask update + AABB shrink;
    the iteration right after the update is checked against the oracle with the same mask and cropped factors."""
    from joint_tensorf_amd.model import tensorf_hip
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views
    B, HW = 3, 40
    opt = make_options("tensorf_blender_VM", device=DEV, model="tensorf_hip", data=dict(image_size=[HW, HW], num_views=B),
                       arch=dict(tensorf=dict(model="TensorVMSplit")),
                       train_schedule=dict(n_voxel_init=16 ** 3, n_rays_init=96, n_rays_rest=96,
                                           update_alphamask_iters=[3, 6], upsample_iters=[100]),
                       nerf=dict(n_rays=96))
    torch.manual_seed(0)
    np.random.seed(0)
    model = tensorf_hip.Model(opt)
    model.build_networks(opt)
    model.setup_optimizer(opt)
    g = model.graph
    tf = g.nerf.tensorf
    assert type(tf).__name__ == "TensorVMSplit"
    with torch.no_grad():  # content concentrated in the middle so that the shrink cuts something away
        for l, pl in zip(tf.density_line, tf.density_plane):
            l.mul_(22.0)
            for t in (l[:, :, :4], l[:, :, 11:], pl[:, :, :4], pl[:, :, 11:], pl[:, :, :, :4], pl[:, :, :, 11:]):
                t.mul_(0.02)
    with torch.no_grad():
        a, _ = tf.getDenseAlpha(tf.gridSize.tolist())
    tf.alphaMask_thres = float(torch.quantile(a.flatten(), 0.9))
    var0 = make_views(opt, B, seed=5, device=DEV)
    grid0 = tf.gridSize.tolist()
    losses = []
    for it in range(5):
        loss = model.train_iteration(opt, Opt(dict(var0)))
        if model.it == 3:  # the update iteration (update_schedule sees the incremented counter)
            with torch.no_grad():
                a, _ = tf.getDenseAlpha(tf.gridSize.tolist())
            tf.alphaMask_thres = float(torch.quantile(a.flatten(), 0.985))
        model.after_iteration(opt)
        losses.append(float(loss.all.detach()))
        assert np.isfinite(losses[-1])
        if it == 2:
            assert tf.alphaMask is not None and tf.gridSize.tolist() != grid0  # mask built, box shrunk
            assert 0.0 < float(tf.alphaMask.alpha_volume.mean()) < 0.9
    # one more forward with fixed draws, against the oracle on the shrunk scene with the same mask
    params = _oracle_params(model)
    cfg = O.SceneCfg(tf.aabb.view(-1).tolist(), tf.gridSize.tolist(), list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio)
    cfg.aabb = tf.aabb.clone()
    mask = (tf.alphaMask.alpha_volume[0, 0].cpu(), tf.alphaMask.aabb.cpu())
    cpu = {k: v.cpu() for k, v in dict(var0).items() if torch.is_tensor(v)}
    ray_idx, _, _, _ = O.rand_grid_ray_idx(HW, HW, 96, B, 1, 2)
    center, ray = O.rays_for_pixels(cpu["pose"], cpu["intr_inv"], ray_idx, HW)
    jit = torch.rand(center.shape[0] * center.shape[1], 1, generator=torch.Generator().manual_seed(5))
    ref = O.render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), g.nerf.n_samples, white_bg=True, jitter=jit,
                   alpha_mask=mask)
    tf.jitter_override = jit.to(DEV)
    try:
        got = tf(opt, center.reshape(-1, 3).to(DEV), ray.reshape(-1, 3).to(DEV), white_bg=True, is_train=True,
                 N_samples=g.nerf.n_samples)
    finally:
        tf.jitter_override = None
    np.testing.assert_allclose(got[0].detach().cpu().numpy(), ref[0].detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(got[2].detach().cpu().numpy(), ref[2].detach().numpy(), atol=2e-5)
