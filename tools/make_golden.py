#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the reference (CPU, this container only).

Runs ONLY where /root/reference exists (the build container).  It imports the reference's
own `model/bat.py` Graph with throw-away stand-ins for its *non-arithmetic* third-party
imports (easydict, icecream, wandb, ...), drives
`Graph.forward -> compute_loss -> backward` on tiny synthetic scenes and stores the captured
inputs, outputs and gradients as small .npz fixtures under tests/golden/.

The fixtures are data only: tensors that went in and tensors that came out.  Nothing of the
reference's source travels.  Re-run:  python tools/make_golden.py  [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


# ----------------------------------------------------------------------------------------------
# stand-ins for the reference's non-arithmetic imports
# ----------------------------------------------------------------------------------------------
class EasyDict(dict):
    """Minimal recursive attribute dict (the reference uses easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = {} if d is None else dict(d)
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _conv(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, list):
            return [cls._conv(e) for e in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._conv(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return a[0] if len(a) == 1 else (a if a else None)

        def __getattr__(self, k):
            return _Anything()

    mod("easydict", EasyDict=EasyDict)
    ic = _Anything()
    mod("icecream", ic=ic)
    mod("wandb", log=lambda *a, **k: None, init=_Anything(), Image=_Anything, Video=_Anything,
        run=_Anything())
    mod("lpips", LPIPS=_Anything)
    mod("visdom", Visdom=_Anything)
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms")
    tvf = mod("torchvision.transforms.functional")
    tv.transforms = tvt
    tvt.functional = tvf
    tv.utils = mod("torchvision.utils", make_grid=_Anything(), save_image=_Anything())
    mod("termcolor", colored=lambda s, **k: str(s))
    mod("ipdb", set_trace=lambda *a, **k: None)
    mod("imageio", imread=_Anything(), imwrite=_Anything())
    tb = mod("torch.utils.tensorboard", SummaryWriter=_Anything)
    torch.utils.tensorboard = tb
    ext = mod("external")
    ssim = mod("external.pohsun_ssim")
    ext.pohsun_ssim = ssim
    ssim.pytorch_ssim = mod("external.pohsun_ssim.pytorch_ssim", ssim=_Anything())


def import_reference():
    _install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)
    import options  # noqa
    import camera  # noqa
    import model.bat as bat  # noqa
    import model.kernels as kernels  # noqa
    return options, camera, bat, kernels


# ----------------------------------------------------------------------------------------------
# synthetic scene
# ----------------------------------------------------------------------------------------------
def look_at_pose(eye):
    """world->camera [R|t] looking at the origin (camera looks along +z, reference convention)."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 0.0, 1.0])
    if abs(fwd @ up) > 0.99:
        up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 0)  # rows = camera axes in world coords
    t = -R @ eye
    return np.concatenate([R, t[:, None]], 1).astype(np.float32)


def make_opt(options, yaml_name, H, W, n_voxel_init, extra=None):
    opt = options.load_options("options/{}.yaml".format(yaml_name))
    over = dict(model="bat", yaml=yaml_name, data=dict(image_size=[H, W]),
                train_schedule=dict(n_voxel_init=n_voxel_init))
    if extra:
        for k, v in extra.items():
            over[k] = v
    opt = options.override_options(opt, EasyDict(over), key_stack=[])
    opt.device = "cpu"
    opt.H, opt.W = H, W
    opt.output_path = "/tmp/jt_golden_out"
    return opt


def build_graph(bat, camera, opt, n_views, seed, se3_scale=0.01, noise=True):
    torch.manual_seed(seed)
    np.random.seed(seed)
    graph = bat.Graph(opt)
    graph.se3_refine = torch.nn.Embedding(n_views, 6)
    with torch.no_grad():
        graph.se3_refine.weight.copy_(torch.randn(n_views, 6) * se3_scale)
    if noise:
        se3_noise = torch.randn(n_views, 6) * float(opt.camera.noise)
        graph.pose_noise = torch.nn.Parameter(camera.lie.se3_to_SE3(se3_noise), requires_grad=False)
    graph.train()
    return graph


def make_var(opt, n_views, seed, llff=False):
    g = torch.Generator().manual_seed(seed + 100)
    H, W = opt.H, opt.W
    if llff:
        # near-identity world->camera poses (LLFF after recentering), looking along +z
        poses = []
        for i in range(n_views):
            eye = np.array([0.15 * np.cos(2.1 * i), 0.12 * np.sin(1.3 * i), 0.02 * i])
            P = np.eye(3, 4, dtype=np.float32)
            P[:, 3] = -eye
            poses.append(P)
        f = 0.8 * W
    else:
        poses = []
        for i in range(n_views):
            th = 2 * np.pi * i / n_views + 0.3
            ph = 0.5 + 0.2 * np.sin(1.7 * i)
            eye = 4.0 * np.array([np.cos(th) * np.cos(ph), np.sin(th) * np.cos(ph), np.sin(ph)])
            poses.append(look_at_pose(eye))
        f = 0.5 * W / np.tan(0.5 * 0.69)
    pose = torch.tensor(np.stack(poses))
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float32)
    intr = intr[None].repeat(n_views, 1, 1)
    var = EasyDict(
        idx=torch.arange(n_views),
        pose=pose,
        intr=intr,
        intr_inv=intr.inverse(),
        image=torch.rand(n_views, 3, H, W, generator=g),
    )
    var.train_edge_masks = (torch.rand(n_views, H * W, generator=g) < 0.5).to(torch.uint8)
    return var


class Recorder:
    """Records the host/device random draws the reference makes on the path."""

    def __init__(self):
        self.rand_like = []
        self.np_choice = []
        self.np_randint = []
        self.coin = []

    def install(self, tensorBase_mod, batBase_mod):
        rec = self
        _rand_like = torch.rand_like
        _choice = np.random.choice
        _randint = np.random.randint
        _rand = torch.rand

        def rand_like(x, *a, **k):
            r = _rand_like(x, *a, **k)
            rec.rand_like.append(r.detach().clone())
            return r

        def choice(pool, *a, **k):
            r = _choice(pool, *a, **k)
            rec.np_choice.append(float(r))
            return r

        def randint(*a, **k):
            r = _randint(*a, **k)
            rec.np_randint.append(int(r))
            return r

        def rand(*a, **k):
            r = _rand(*a, **k)
            if tuple(r.shape) == (1,):
                rec.coin.append(float(r))
            return r

        torch.rand_like = rand_like
        np.random.choice = choice
        np.random.randint = randint
        torch.rand = rand
        self._restore = (_rand_like, _choice, _randint, _rand)

    def uninstall(self):
        torch.rand_like, np.random.choice, np.random.randint, torch.rand = self._restore


def state_np(graph):
    return {"param." + k: v.detach().cpu().numpy().copy() for k, v in graph.state_dict().items()}


def run_case(bat, camera, opt, graph, var, mode, it, progress, out_path, llff=False, extra_meta=None,
             torch_seed=None, extra_arrays=None):
    import model.tensorf_repr.tensorBase as tB
    import model.tensorf_repr.batBase as bB
    import util

    if torch_seed is not None:
        torch.manual_seed(torch_seed)
    graph.it = it
    graph.nerf.progress.data.fill_(progress)
    for p in graph.parameters():
        p.grad = None
    rec = Recorder()
    rec.install(tB, bB)
    # capture what tensorf.forward is called with (blur params after random scale/cut-off etc.)
    tf = graph.nerf.tensorf
    captured = {}
    orig_forward = tf.forward

    def forward_spy(opt_, **kw):
        captured.update({k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
        captured["near_far"] = [float(tf.near_far[0]), float(tf.near_far[1])]
        out = orig_forward(opt_, **kw)
        captured["kernel_density"] = None if tf.kernel_density is None else tf.kernel_density.detach().clone()
        captured["kernel_color"] = None if tf.kernel_color is None else tf.kernel_color.detach().clone()
        return out

    tf.forward = forward_spy
    try:
        v = EasyDict(var)
        if mode == "vis":
            # deterministic path used by render_by_slices (nerf.py:728-740): one slice of ray_idx
            pose = graph.get_pose(opt, v, mode="train")
            ray_idx = torch.arange(0, opt.H * opt.W, 7)[:24]
            ret = graph.render(opt, pose, intr_inv=v.intr_inv, ray_idx=ray_idx, mode="vis", intr=v.intr)
            v.update(ret)
            v.ray_idx = ray_idx
            v.current_pose = pose
            loss_in = v.rgb
            loss = EasyDict(render=((v.rgb - 0.3) ** 2).mean())
            loss_all = loss.render
        else:
            v = graph.forward(opt, v, mode=mode)
            loss = graph.compute_loss(opt, v, mode=mode)
            loss_all = 0.0
            w_l1 = float(opt.loss_weight.L1.init)
            for k in loss:
                if k == "L1":
                    loss_all = loss_all + w_l1 * loss[k]
                elif opt.loss_weight[k] is not None:
                    loss_all = loss_all + float(opt.loss_weight[k]) * loss[k]
        loss_all.backward()
    finally:
        rec.uninstall()
        tf.forward = orig_forward

    out = {}
    out.update(state_np(graph))
    for k, p in graph.named_parameters():
        if p.grad is not None:
            out["grad." + k] = p.grad.detach().cpu().numpy().copy()
    out["in.pose_gt"] = var.pose.numpy()
    out["in.intr"] = var.intr.numpy()
    out["in.intr_inv"] = var.intr_inv.numpy()
    out["in.idx"] = var.idx.numpy()
    out["in.image"] = var.image.numpy()
    out["in.train_edge_masks"] = var.train_edge_masks.numpy()
    out["in.ray_idx"] = v.ray_idx.numpy()
    out["mid.current_pose"] = v.current_pose.detach().numpy()
    out["mid.center"] = captured["center"].numpy()
    out["mid.ray_dir"] = captured["ray_dir"].numpy()
    if rec.rand_like:
        out["in.jitter"] = rec.rand_like[0].numpy()
    out["out.rgb"] = v.rgb.detach().numpy()
    out["out.depth"] = v.depth.detach().numpy()
    out["out.opacity"] = v.opacity.detach().numpy()
    for k in loss:
        out["loss." + k] = np.float32(loss[k].detach().item() if isinstance(loss[k], torch.Tensor) else loss[k])
    out["loss.all"] = np.float32(loss_all.detach().item())
    if captured["kernel_density"] is not None:
        out["mid.kernel_density"] = captured["kernel_density"].numpy()
        out["mid.kernel_color"] = captured["kernel_color"].numpy()
    meta = dict(
        mode=mode, it=it, progress=float(progress), H=opt.H, W=opt.W, llff=bool(llff),
        yaml=str(opt.yaml),
        white_bg=bool(captured["white_bg"]), is_train=bool(captured["is_train"]),
        ndc_ray=bool(captured["ndc_ray"]), N_samples=int(captured["N_samples"]),
        c2f_parameter_density=None if captured["c2f_parameter_density"] is None else float(captured["c2f_parameter_density"]),
        c2f_parameter_color=None if captured["c2f_parameter_color"] is None else float(captured["c2f_parameter_color"]),
        c2f_mode=captured["c2f_mode"], c2f_kernel_size=captured["c2f_kernel_size"],
        fea_pe_progress=float(captured["fea_pe_progress"]), view_pe_progress=float(captured["view_pe_progress"]),
        is_test_optim=bool(captured["is_test_optim"]),
        near_far=captured["near_far"],
        coin=rec.coin, np_choice=rec.np_choice, np_randint=rec.np_randint,
        gridSize=[int(x) for x in tf.gridSize.tolist()],
        aabb=[float(x) for x in tf.aabb.view(-1).tolist()],
        stepSize=float(tf.stepSize), step_ratio=float(tf.step_ratio),
        density_shift=float(tf.density_shift), distance_scale=float(tf.distance_scale),
        fea2denseAct=str(tf.fea2denseAct), rayMarch_weight_thres=float(tf.rayMarch_weight_thres),
        shadingMode=str(tf.shadingMode), view_pe=int(tf.view_pe), fea_pe=int(tf.fea_pe),
        density_n_comp=[int(x) for x in tf.density_n_comp], app_n_comp=[int(x) for x in tf.app_n_comp],
        app_dim=int(tf.app_dim), featureC=int(tf.featureC),
        L1_weight=float(opt.loss_weight.L1.init),
        TV_density_weight=float(opt.loss_weight.TV_density), TV_color_weight=float(opt.loss_weight.TV_color),
        ndc_near_plane=float(opt.arch.ndc_near_plane) if hasattr(opt.arch, "ndc_near_plane") else 1.0,
        edge_loss=dict(on=bool(getattr(opt, "edge_mask_on_render_loss", False)),
                       alternate=bool(getattr(opt, "alternate_edge_loss", False)),
                       before_iter=int(getattr(opt, "edge_mask_before_iter", 0)),
                       edge_factor=float(getattr(opt, "edge_loss_factor", 1.0)),
                       non_edge_factor=float(getattr(opt, "non_edge_loss_factor", 1.0))),
        ray_sampling_strategy=str(opt.nerf.ray_sampling_strategy),
        grid_H=int(v.grid_H) if "grid_H" in v else -1, grid_W=int(v.grid_W) if "grid_W" in v else -1,
    )
    if extra_meta:
        meta.update(extra_meta)
    if extra_arrays:
        out.update(extra_arrays)
    import json
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "loss.all=%.6f" % out["loss.all"], "R=%d S=%d" % (out["out.rgb"].shape[0] * out["out.rgb"].shape[1], meta["N_samples"]),
          "size=%.0f KB" % (os.path.getsize(out_path) / 1024))


def eval_case(bat, camera, opt, graph, var, out_path, test_iter=4, seed=21):
    """N1: Procrustes pre-alignment of the optimised training cameras, a short test-time
    photometric pose optimisation of one held-out view and the sliced full-image render + PSNR, all run by
    the reference's own bat.Model methods (model/bat.py:205-292, model/nerf.py:525-548)."""
    import types
    import json
    import model.tensorf as ref_tensorf
    import model.tensorf_repr.tensorBase as tB
    import model.tensorf_repr.batBase as bB

    torch.manual_seed(seed)
    np.random.seed(seed)
    opt.optim.test_iter = test_iter
    graph.eval()
    graph.it = 5
    graph.nerf.progress.data.fill_(0.9)
    n_views = var.pose.shape[0]
    # all training poses as bat.Model.get_all_training_poses composes them (model/bat.py:197-210)
    with torch.no_grad():
        pose_GT = var.pose
        pose = camera.pose.compose([graph.pose_noise, pose_GT])
        pose = camera.pose.compose([camera.lie.se3_to_SE3(graph.se3_refine.weight), pose])
    me = types.SimpleNamespace(graph=graph, it=graph.it)
    me.summarize_loss = types.MethodType(ref_tensorf.Model.summarize_loss, me)
    pose_aligned, sim3 = bat.Model.prealign_cameras(me, opt, pose, pose_GT)
    err = bat.Model.evaluate_camera_alignment(me, opt, pose_aligned, pose_GT)
    graph.sim3 = sim3
    # The three fixture cameras are coplanar once centred, so the rotation above is decided by round-off in its
    # null direction: fine as an INPUT of the test-time optimisation below, useless as a known answer.  The
    # known answer for the alignment itself uses seven cameras at different heights.
    ga = torch.Generator().manual_seed(seed + 7)
    pgt7 = torch.tensor(np.stack([look_at_pose(4.0 * np.array([np.cos(0.9 * i) * np.cos(0.25 * i - 0.6),
                                                                np.sin(0.9 * i) * np.cos(0.25 * i - 0.6),
                                                                np.sin(0.25 * i - 0.6)])) for i in range(7)]))
    with torch.no_grad():
        p7 = camera.pose.compose([camera.lie.se3_to_SE3(torch.randn(7, 6, generator=ga) * 0.15), pgt7])
        # a global similarity on top, as joint optimisation leaves the scene in its own gauge
        gR = camera.lie.se3_to_SE3(torch.tensor([[0.3, -0.2, 0.5, 0.0, 0.0, 0.0]]))[0, :, :3]
        c7 = camera.cam2world(torch.zeros(1, 1, 3), p7)[:, 0] * 1.3 @ gR.t() + torch.tensor([0.2, -0.1, 0.3])
        R7 = p7[..., :3] @ gR.t()
        p7 = camera.pose(R=R7, t=(-R7 @ c7[..., None])[..., 0])
    aligned7, sim7 = bat.Model.prealign_cameras(me, opt, p7, pgt7)
    err7 = bat.Model.evaluate_camera_alignment(me, opt, aligned7, pgt7)
    # one held-out view
    g = torch.Generator().manual_seed(seed + 1)
    eye = 4.0 * np.array([np.cos(1.1) * np.cos(0.6), np.sin(1.1) * np.cos(0.6), np.sin(0.6)])
    tvar = EasyDict(idx=torch.arange(1), pose=torch.tensor(look_at_pose(eye))[None], intr=var.intr[:1],
                    intr_inv=var.intr_inv[:1], image=torch.rand(1, 3, opt.H, opt.W, generator=g))
    rec = Recorder()
    rec.install(tB, bB)
    trace_se3, trace_loss = [], []
    orig_compute = graph.compute_loss

    def compute_spy(opt_, v, mode=None):
        out = orig_compute(opt_, v, mode=mode)
        if mode == "test-optim":
            trace_se3.append(v.se3_refine_test.detach().clone())
            trace_loss.append(float(out.render.detach()))
        return out

    graph.compute_loss = compute_spy
    tf = graph.nerf.tensorf
    calls = []
    orig_forward = tf.forward

    def forward_spy(opt_, **kw):
        calls.append({k: (None if v is None else (v if isinstance(v, (bool, int, float, str)) else None))
                      for k, v in kw.items() if not isinstance(v, torch.Tensor)})
        return orig_forward(opt_, **kw)

    tf.forward = forward_spy
    try:
        v = bat.Model.evaluate_test_time_photometric_optim(me, opt, EasyDict(tvar))
        with torch.no_grad():
            v = graph.forward(opt, v, mode="eval")
            rgb_map = v.rgb.view(-1, opt.H, opt.W, 3).permute(0, 3, 1, 2)
            psnr = -10 * graph.MSE_loss(rgb_map, v.image).log10().item()
    finally:
        rec.uninstall()
        graph.compute_loss = orig_compute
        tf.forward = orig_forward
    out = {}
    out.update(state_np(graph))
    out["in.pose_gt"] = var.pose.numpy()
    out["in.test_pose"] = tvar.pose.numpy()
    out["in.test_image"] = tvar.image.numpy()
    out["in.intr"] = tvar.intr.numpy()
    out["in.intr_inv"] = tvar.intr_inv.numpy()
    out["mid.pose_all"] = pose.numpy()
    out["mid.pose_aligned"] = pose_aligned.numpy()
    out["sim3.t0"] = sim3.t0.numpy()
    out["sim3.t1"] = sim3.t1.numpy()
    out["sim3.s0"] = np.float32(sim3.s0)
    out["sim3.s1"] = np.float32(sim3.s1)
    out["sim3.R"] = sim3.R.numpy()
    out["err.R"] = err.R.numpy()
    out["err.t"] = err.t.numpy()
    out["align.pose"] = p7.numpy()
    out["align.pose_gt"] = pgt7.numpy()
    out["align.pose_aligned"] = aligned7.numpy()
    out["align.sim3.t0"] = sim7.t0.numpy()
    out["align.sim3.t1"] = sim7.t1.numpy()
    out["align.sim3.s0"] = np.float32(sim7.s0)
    out["align.sim3.s1"] = np.float32(sim7.s1)
    out["align.sim3.R"] = sim7.R.numpy()
    out["align.err.R"] = err7.R.numpy()
    out["align.err.t"] = err7.t.numpy()
    out["trace.se3"] = torch.stack(trace_se3).numpy()      # value BEFORE the step of iteration k
    out["trace.loss_render"] = np.array(trace_loss, dtype=np.float32)
    out["out.se3_refine_test"] = v.se3_refine_test.detach().numpy()
    out["out.rgb"] = v.rgb.numpy()
    out["out.depth"] = v.depth.numpy()
    out["out.opacity"] = v.opacity.numpy()
    out["out.psnr"] = np.float32(psnr)
    meta = dict(test_iter=test_iter, H=opt.H, W=opt.W, it=graph.it, progress=0.9, n_rays=int(opt.nerf.n_rays),
                forward_calls_optim=calls[:test_iter], forward_call_eval=calls[-1], n_forward_calls=len(calls),
                aabb=[float(x) for x in tf.aabb.view(-1).tolist()], near_far=[float(tf.near_far[0]), float(tf.near_far[1])],
                stepSize=float(tf.stepSize), step_ratio=float(tf.step_ratio), density_shift=float(tf.density_shift),
                distance_scale=float(tf.distance_scale), fea2denseAct=str(tf.fea2denseAct),
                rayMarch_weight_thres=float(tf.rayMarch_weight_thres), shadingMode=str(tf.shadingMode),
                view_pe=int(tf.view_pe), fea_pe=int(tf.fea_pe), ndc_near_plane=1.0, llff=False,
                TV_density_weight=float(opt.loss_weight.TV_density), TV_color_weight=float(opt.loss_weight.TV_color),
                test_photo=bool(opt.optim.test_photo),
                lr_pose=float(opt.optim.lr_pose), lr_pose_test=float(opt.optim.lr_pose_test),
                lr_pose_test_end=float(opt.optim.lr_pose_test_end), np_randint=rec.np_randint, np_choice=rec.np_choice,
                ray_sampling_strategy=str(opt.nerf.ray_sampling_strategy), yaml=str(opt.yaml),
                gridSize=[int(x) for x in tf.gridSize.tolist()], N_samples=int(graph.nerf.n_samples),
                L1_weight=float(opt.loss_weight.L1.init))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "psnr=%.3f" % psnr, "se3_test=", v.se3_refine_test.detach().numpy().round(6).tolist(),
          "loss trace", [round(x, 6) for x in trace_loss])


def gt_blur_case(bat, opt, out_path, n_views=5, seed=31):
    """N3: the 2-D blur cache of the supervising images and the Sobel edge masks, by the
    reference's own nerf.Model.process_GT_images / get_edge_mask (model/nerf.py:57-149)."""
    import types
    import json
    import model.nerf as ref_nerf
    import util_vis
    g = torch.Generator().manual_seed(seed)
    H, W = opt.H, opt.W
    # smooth random images with some structure (pure noise has no meaningful edges)
    base = torch.rand(n_views, 3, H // 4, W // 4, generator=g)
    images = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=True)
    images = (images + 0.15 * torch.rand(n_views, 3, H, W, generator=g)).clamp(0, 1).contiguous()
    me = types.SimpleNamespace(it=700, tb=None)
    me.train_data = types.SimpleNamespace(all=EasyDict(image=images))
    orig = util_vis.tb_wandb_image
    util_vis.tb_wandb_image = lambda *a, **k: None
    try:
        blurred = ref_nerf.Model.process_GT_images(me, opt)
        masks = ref_nerf.Model.get_edge_mask(me, opt, blurred)
    finally:
        util_vis.tb_wandb_image = orig
    out = {"in.images": images.numpy()}
    scales = sorted(blurred.keys())
    for sc in scales:
        out["blur.%g" % sc] = blurred[sc].numpy()
        out["mask.%g" % sc] = masks[sc].numpy()
    meta = dict(it=700, max_iter=int(opt.max_iter), H=H, W=W, scales=[float(s) for s in scales],
                blur_2d_c2f_schedule=[float(x) for x in opt.blur_2d_c2f_schedule],
                blur_2d_c2f_kernel_size=int(opt.blur_2d_c2f_kernel_size), blur_2d_mode=str(opt.blur_2d_mode),
                hard_edge_mask_mean_thresh=float(opt.hard_edge_mask_mean_thresh), soft_edge_mask=bool(opt.soft_edge_mask))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "scales", scales, "mask fractions", [float(masks[s].float().mean()) for s in scales])


def mask_case(bat, camera, opt, graph, var, outdir, grid=(20, 20, 20), quantile=0.93):
    """N4: alpha-mask volume, masked render and AABB shrink by the reference's own
    TensorBase.updateAlphaMask / BatBase.forward / TensorVMSplit.shrink (tensorBase.py:618-661, batBase.py:76-82,
    tensoRF.py:297-334).  The threshold is set so that the mask is non-trivial on this random field."""
    tf = graph.nerf.tensorf
    tf.kernel_density, tf.c2f_mode = None, None
    with torch.no_grad():
        # concentrate the content in the middle of the box so that the shrink has something to cut away
        for l, pl in zip(tf.density_line, tf.density_plane):
            l[:, :, :4] *= 0.02
            l[:, :, 10:] *= 0.02
            pl[:, :, :4] *= 0.02
            pl[:, :, 10:] *= 0.02
            pl[:, :, :, :4] *= 0.02
            pl[:, :, :, 10:] *= 0.02
        alpha, dense_xyz = tf.getDenseAlpha(grid)
    tf.alphaMask_thres = float(torch.quantile(alpha.flatten(), quantile))
    pre = dict(aabb=tf.aabb.clone(), gridSize=tf.gridSize.clone())
    new_aabb = tf.updateAlphaMask(grid)
    vol = tf.alphaMask.alpha_volume
    print("alpha mask: thres %.3e, kept %.1f %%, new aabb %s" % (tf.alphaMask_thres, 100 * float(vol.mean()), new_aabb.tolist()))
    arrays = {"mask.alpha_volume": vol[0, 0].numpy().copy(), "mask.aabb": tf.alphaMask.aabb.numpy().copy(),
              "mask.dense_alpha": alpha.numpy().copy(), "mask.new_aabb": new_aabb.numpy().copy(),
              "mask.thres": np.float32(tf.alphaMask_thres), "mask.grid": np.array(grid, np.int32)}
    run_case(bat, camera, opt, graph, var, "train", it=8, progress=0.9,
             out_path=os.path.join(outdir, "blender_train_alphamask.npz"), extra_arrays=arrays)
    tf.shrink(new_aabb)
    arrays2 = dict(arrays)
    arrays2["shrink.aabb_before"] = pre["aabb"].numpy()
    arrays2["shrink.gridSize_before"] = pre["gridSize"].numpy()
    run_case(bat, camera, opt, graph, var, "train", it=10, progress=0.9,
             out_path=os.path.join(outdir, "blender_train_shrunk.npz"), extra_arrays=arrays2)


def known_answers(camera, kernels, bat, out_path):
    """Known-answer vectors for the small pure functions on the path."""
    import model.tensorf_repr.bateRF as bateRF
    torch.manual_seed(7)
    out = {}
    wu = torch.cat([torch.randn(6, 6) * 0.3, torch.zeros(1, 6), torch.randn(2, 6) * 1e-4,
                    torch.tensor([[3.0, 0.1, -0.2, 0.5, -0.4, 0.3]])], 0).requires_grad_(True)
    Rt = camera.lie.se3_to_SE3(wu)
    cot = torch.randn(Rt.shape)
    (Rt * cot).sum().backward()
    out["se3.wu"] = wu.detach().numpy()
    out["se3.Rt"] = Rt.detach().numpy()
    out["se3.cot"] = cot.numpy()
    out["se3.grad_wu"] = wu.grad.numpy()
    pa, pb = torch.randn(5, 3, 4), torch.randn(5, 3, 4)
    out["compose.a"] = pa.numpy()
    out["compose.b"] = pb.numpy()
    out["compose.ab"] = camera.pose.compose_pair(pa, pb).numpy()
    sig = [1e-5, 0.05, 0.37, 1.0, 2.68, 4.8, 6.4, 7.94]
    out["gauss.sigma"] = np.array(sig, np.float32)
    out["gauss.k65"] = np.stack([kernels.get_gaussian_kernel(torch.tensor(s), 64).numpy() for s in sig])
    out["gauss.k9"] = np.stack([kernels.get_gaussian_kernel(torch.tensor(s), 8).numpy() for s in sig])
    # separable blur of planes/lines through the reference's own methods (cubic and non-cubic)
    conv_plane = bateRF.BAT_VMSplit.convolute_plane
    conv_line = bateRF.BAT_VMSplit.convolute_line
    k = kernels.get_gaussian_kernel(torch.tensor(2.3), 16).view(1, 1, -1)
    out["blur.kernel"] = k.view(-1).numpy()
    for name, (C, gm0, gm1) in dict(cubic=(5, 11, 11), noncubic=(4, 9, 13)).items():
        plane = torch.randn(1, C, gm1, gm0)  # [1,C,g[m1],g[m0]] as in tensoRF.py:165
        out["blur.%s.in" % name] = plane.numpy()
        # reference call site passes (H,W) = (g[m0], g[m1])  (bateRF.py:68,76)
        out["blur.%s.out" % name] = conv_plane(None, k, plane, gm0, gm1).numpy()
    line = torch.randn(1, 6, 17, 1)
    out["blur.line.in"] = line.numpy()
    out["blur.line.out"] = conv_line(None, k, line).numpy()
    # grid upsampling of the VM factors (tensoRF.py:274-295), non-cubic 6x7x5 -> 9x11x8
    import types
    import model.tensorf_repr.tensoRF as tensoRF
    vm = types.SimpleNamespace(matMode=[[0, 1], [0, 2], [1, 2]], vecMode=[2, 1, 0])
    g0, g1 = [6, 7, 5], [9, 11, 8]
    planes = [torch.randn(1, 4, g0[m1], g0[m0]) for m0, m1 in vm.matMode]
    lines = [torch.randn(1, 4, g0[v], 1) for v in vm.vecMode]
    for i in range(3):
        out["up.plane_in.%d" % i] = planes[i].numpy()
        out["up.line_in.%d" % i] = lines[i].numpy()
    up_p, up_l = tensoRF.TensorVMSplit.up_sampling_VM(vm, [torch.nn.Parameter(p.clone()) for p in planes],
                                                      [torch.nn.Parameter(l.clone()) for l in lines], g1)
    for i in range(3):
        out["up.plane_out.%d" % i] = up_p[i].detach().numpy()
        out["up.line_out.%d" % i] = up_l[i].detach().numpy()
    out["up.res_target"] = np.array(g1, np.int32)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path)


def ckpt_case(bat, camera, options, out_path, seed=41):
    """A checkpoint as the reference's own util.save_checkpoint writes it (util.py:162-184), taken AFTER a grid
    upsampling (so that its tensors no longer fit a freshly built model) and two optimizer steps, unpacked into plain
    arrays + JSON: graph state_dict, manually tracked parameters, Adam state; plus what the reference renders from the
    restored state (mode "vis") -- the target of the build's restore_checkpoint."""
    import json
    import shutil
    import util
    B = 3
    opt = make_opt(options, "bat_blender_VM", H=32, W=32, n_voxel_init=10 ** 3,
                   extra=dict(nerf=dict(n_rays=60)))
    opt.train_schedule.n_voxel_final, opt.train_schedule.upsample_iters = 16 ** 3, [2, 50]
    opt.output_path = "/tmp/jt_golden_ckpt"
    shutil.rmtree(opt.output_path, ignore_errors=True)
    graph = build_graph(bat, camera, opt, B, seed=seed)
    var = make_var(opt, B, seed=seed)
    nerf = graph.nerf

    class M:
        pass
    m = M()
    m.graph = graph
    m.train_data = list(range(B))
    m.optim = nerf._get_optimizer(opt)
    nerf.get_current_optimizer = lambda: m.optim

    def register(o):
        m.optim = o
    nerf.register_new_optimizer = register
    m.save_param_state = lambda: graph.save_param_state()
    with torch.no_grad():
        for p in nerf.tensorf.density_plane:
            p.mul_(22.0)
    rs = np.random.RandomState(seed)
    for it in range(1, 5):  # iterations 1..4: the upsampling happens at update_schedule(opt, 2)
        graph.it = it
        nerf.progress.data.fill_(it / opt.max_iter)
        m.optim.zero_grad()
        np.random.seed(int(rs.randint(1 << 30)))
        torch.manual_seed(int(rs.randint(1 << 30)))
        v = graph.forward(opt, EasyDict(var), mode="train")
        loss = graph.compute_loss(opt, v, mode="train")
        (loss.render + float(opt.loss_weight.L1.init) * loss.L1).backward()
        m.optim.step()
        nerf.update_schedule(opt, it)
    it_saved = 4
    util.save_checkpoint(opt, m, ep=None, it=it_saved, latest=True)
    ck = torch.load("{}/model.ckpt".format(opt.output_path), map_location="cpu", weights_only=False)
    out = {}
    for k, t in ck["graph"].items():
        out["graph." + k] = t.detach().cpu().numpy()
    mt = ck["manually_tracked_parameters"]
    trk = {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in mt["tensorf_reset_kwargs"].items()}
    nrk = {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in mt["nerf_reset_kwargs"].items()}
    groups = []
    for gi, g in enumerate(ck["optim"]["param_groups"]):
        groups.append({k: v for k, v in g.items() if k != "params"} | {"params": list(g["params"])})
    for pid, st in ck["optim"]["state"].items():
        out["optim.state.%d.step" % pid] = np.float32(float(st["step"]))
        out["optim.state.%d.exp_avg" % pid] = st["exp_avg"].numpy()
        out["optim.state.%d.exp_avg_sq" % pid] = st["exp_avg_sq"].numpy()
    # what the reference renders after restoring this file into a FRESH model (the path train_3d.py:95-103 takes)
    opt2 = make_opt(options, "bat_blender_VM", H=32, W=32, n_voxel_init=10 ** 3,
                    extra=dict(nerf=dict(n_rays=60)))
    opt2.train_schedule.n_voxel_final, opt2.train_schedule.upsample_iters = 16 ** 3, [2, 50]
    opt2.output_path = opt.output_path
    graph2 = build_graph(bat, camera, opt2, B, seed=seed + 1)
    m2 = M()
    m2.graph = graph2
    m2.train_data = list(range(B))
    m2.load_param_state = lambda o, c: graph2.load_param_state(o, c)
    _load = torch.load  # the reference targets torch 1.13, where torch.load unpickles arbitrary objects (its `opt`)
    torch.load = lambda *a, **k: _load(*a, **{**k, "weights_only": False})
    try:
        util.restore_checkpoint(opt2, m2, load_name="{}/model.ckpt".format(opt.output_path))
    finally:
        torch.load = _load
    graph2.eval()
    with torch.no_grad():
        pose = graph2.get_pose(opt2, EasyDict(var), mode="train")
        ray_idx = torch.arange(0, opt2.H * opt2.W, 5)[:40]
        ret = graph2.render(opt2, pose, intr_inv=var.intr_inv, ray_idx=ray_idx, mode="vis", intr=var.intr)
    out["in.pose_gt"], out["in.intr"], out["in.intr_inv"] = var.pose.numpy(), var.intr.numpy(), var.intr_inv.numpy()
    out["in.ray_idx"] = ray_idx.numpy()
    out["out.rgb"], out["out.depth"], out["out.opacity"] = (ret[k].numpy() for k in ("rgb", "depth", "opacity"))
    out["out.current_pose"] = pose.numpy()
    meta = dict(iter=it_saved, epoch=None, tensorf_reset_kwargs=trk, nerf_reset_kwargs=nrk, optim_param_groups=groups,
                gridSize=graph2.nerf.tensorf.gridSize.tolist(), n_samples=int(graph2.nerf.n_samples), H=opt.H, W=opt.W,
                n_views=B, progress=float(graph2.nerf.progress),
                overrides=dict(image_size=[32, 32], n_voxel_init=10 ** 3, n_voxel_final=16 ** 3, upsample_iters=[2, 50],
                               n_rays=60))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "grid", meta["gridSize"], "lr", [g["lr"] for g in groups])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    args = ap.parse_args()
    outdir = os.path.abspath(args.out)
    os.makedirs(outdir, exist_ok=True)
    options, camera, bat, kernels = import_reference()
    torch.set_num_threads(4)

    known_answers(camera, kernels, bat, os.path.join(outdir, "known_answers.npz"))
    ckpt_case(bat, camera, options, os.path.join(outdir, "reference_checkpoint.npz"))

    # ---- Blender (bat_blender_VM): cubic grid, MLP_Fea, softplus, white bg -----------------------
    B = 3
    opt = make_opt(options, "bat_blender_VM", H=40, W=40, n_voxel_init=14 ** 3,
                   extra=dict(nerf=dict(n_rays=96)))
    graph = build_graph(bat, camera, opt, B, seed=1)
    var = make_var(opt, B, seed=1)
    run_case(bat, camera, opt, graph, var, "train", it=0, progress=0.0,
             out_path=os.path.join(outdir, "blender_train_blur.npz"))
    run_case(bat, camera, opt, graph, var, "train", it=1, progress=0.9,
             out_path=os.path.join(outdir, "blender_train_sharp.npz"))
    run_case(bat, camera, opt, graph, var, "vis", it=2, progress=0.05,
             out_path=os.path.join(outdir, "blender_vis_blur.npz"))
    # semi-transparent content (acc ~ 0.3-0.9): the numerically most informative case
    with torch.no_grad():
        for p in graph.nerf.tensorf.density_plane:
            p.mul_(22.0)
    run_case(bat, camera, opt, graph, var, "train", it=6, progress=0.9,
             out_path=os.path.join(outdir, "blender_train_mid.npz"))
    run_case(bat, camera, opt, graph, var, "train", it=7, progress=0.2,
             out_path=os.path.join(outdir, "blender_train_mid_blur.npz"))
    # denser content: acc ~ 1, part of the samples fall under rayMarch_weight_thres
    with torch.no_grad():
        for p in graph.nerf.tensorf.density_plane:
            p.mul_(40.0 / 22.0)
    run_case(bat, camera, opt, graph, var, "train", it=3, progress=0.9,
             out_path=os.path.join(outdir, "blender_train_dense.npz"))
    run_case(bat, camera, opt, graph, var, "train", it=4, progress=0.1,
             out_path=os.path.join(outdir, "blender_train_dense_blur.npz"))
    # N1: camera pre-alignment, test-time pose optimisation and the sliced eval render (semi-transparent field)
    with torch.no_grad():
        for p in graph.nerf.tensorf.density_plane:
            p.mul_(22.0 / 40.0)
    eval_case(bat, camera, opt, graph, var, os.path.join(outdir, "blender_test_optim.npz"))
    # N3: 2-D blur cache of the GT images + edge masks (host torch ops in the reference, every 500 iterations)
    gt_blur_case(bat, opt, os.path.join(outdir, "gt_blur_edge.npz"))
    # N4: alpha-mask volume, masked render, AABB shrink (last: it changes the scene's box and grid)
    graph.train()
    mask_case(bat, camera, opt, graph, var, outdir)

    # all_view_rand_rays variant (config C1 uses it)
    opt2 = make_opt(options, "bat_blender_VM", H=40, W=40, n_voxel_init=14 ** 3,
                    extra=dict(nerf=dict(n_rays=60, ray_sampling_strategy="all_view_rand_rays")))
    graph2 = build_graph(bat, camera, opt2, B, seed=2)
    var2 = make_var(opt2, B, seed=2)
    run_case(bat, camera, opt2, graph2, var2, "train", it=5, progress=0.5,
             out_path=os.path.join(outdir, "blender_train_randrays.npz"))

    # ---- LLFF (bat_llff_VM_MLP): NDC rays, non-cubic grid, WeakView MLP, relu ---------------------
    Bl = 3
    optl = make_opt(options, "bat_llff_VM_MLP", H=30, W=40, n_voxel_init=2200,
                    extra=dict(nerf=dict(n_rays=90)))
    optl.train_schedule.n_rays_init = 90
    graphl = build_graph(bat, camera, optl, Bl, seed=3, noise=False)
    varl = make_var(optl, Bl, seed=3, llff=True)
    # give the field some content so relu density is non-trivial
    run_case(bat, camera, optl, graphl, varl, "train", it=0, progress=0.9,
             out_path=os.path.join(outdir, "llff_train_sharp.npz"), llff=True)
    run_case(bat, camera, optl, graphl, varl, "train", it=1, progress=0.05,
             out_path=os.path.join(outdir, "llff_train_blur.npz"), llff=True)
    # thinner content + a seed whose CPU coin (batBase.py:154) comes up < 0.5 -> white background added
    with torch.no_grad():
        for p in graphl.nerf.tensorf.density_plane:
            p.mul_(0.02)
    run_case(bat, camera, optl, graphl, varl, "train", it=2, progress=0.12,
             out_path=os.path.join(outdir, "llff_train_thin_whitebg.npz"), llff=True, torch_seed=11)


if __name__ == "__main__":
    main()
