#!/usr/bin/env python3
"""Engine trace: the reference's OWN training loop, recorded (CPU, this container only).

tools/make_golden.py pins the renderer (one forward / backward per fixture).  This script pins the ENGINE around it:
it builds the reference's `model.bat.Model` on a tiny synthetic scene, shortens the yaml's schedule so that every
event of the real one happens inside a few dozen iterations (all four grid upsamplings, the ray-count switch, the
pose reset, pose-gradient accumulation 8 -> 1, the pose-lr warm-up, the end of the factor blur, the edge-loss horizon,
the L1 init -> rest switch, an alpha-mask update) and runs `Model.train(opt)` itself -- model/nerf.py:150-278 driving
model/bat.py:96-116, model/base.py:154-172 and model/tensorf.py:399-447 -- with recorders around it.  Per iteration
it stores what the engine decided (every param-group lr, the pose lr as the pose step saw it, `progress`, the ray
count, the accumulation period, whether `optim_pose.step()` fired, the keyword arguments that reached
`tensorf.forward`, the loss weights, the 2-D supervision scale) and what came out (loss terms, the pose parameters);
and the host / device random draws in the order the loop consumed them, so that the build's loop can be replayed on
the same draws (tests/test_engine_trace.py).

Data only: tensors and numbers.  Re-run:  python tools/make_engine_trace.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402  (the stand-ins for the reference's non-arithmetic imports, the synthetic views)

EasyDict = MG.EasyDict


# what is shortened, per yaml: (overrides of make_opt, n_views, H, W)
CASES = {
    "blender": dict(
        yaml="bat_blender_VM", n_views=5, H=32, W=32, llff=False,
        over=dict(max_iter=48, edge_mask_before_iter=24, nerf=dict(n_rays=80),
                  train_schedule=dict(n_voxel_init=10 ** 3, n_voxel_final=22 ** 3, upsample_iters=[6, 14, 20, 26],
                                      update_alphamask_iters=[30, 44], change_n_rays_after_n_iters=18, n_rays_init=80,
                                      n_rays_rest=60)),
        density_scale=22.0),
    "llff": dict(
        yaml="bat_llff_VM_MLP", n_views=6, H=30, W=40, llff=True,
        over=dict(max_iter=64, edge_mask_before_iter=20, nerf=dict(n_rays=168),
                  optim=dict(warmup_pose=10),
                  train_schedule=dict(n_voxel_init=10 ** 3, n_voxel_final=24 ** 3, upsample_iters=[8, 16, 30, 36],
                                      update_alphamask_iters=[34], reset_pose_on_iter=12,
                                      change_n_rays_after_n_iters=8, n_rays_init=168, n_rays_rest=84,
                                      change_n_AccumPoseGrad_after_n_iters=36, n_AccumPoseGrad_init=8,
                                      n_AccumPoseGrad_rest=1)),
        density_scale=4.0),
}


# renderer fixtures taken inside the reference's loop (tests/golden_util.py CASES): LLFF iteration 21 = the second iteration
# with a NEGATIVE near plane (the schedule passes through zero between 18 and 19), sharp, 14 x 16 x 14 grid after two upsamplings;
# iteration 40 = final-but-one grid behind the alpha-mask update, pose steps every iteration
SNAPSHOTS = ["llff:21", "llff:40", "blender:33"]


class FakeTrainData:
    """what bat.Model.build_networks / nerf.Model.train read of a dataset: len(), .all"""

    def __init__(self, var):
        self.all = var

    def __len__(self):
        return int(self.all.idx.shape[0])


def scalar_kwargs(kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            out[k + ".shape"] = list(v.shape)
        elif v is None or isinstance(v, (bool, int, float, str)):
            out[k] = v
        else:
            try:
                out[k] = float(v)
            except Exception:
                out[k] = str(v)
    return out


def build_reference_model(case_name):
    """the reference's bat.Model on the case's tiny scene, ready for Model.train(opt)"""
    case = CASES[case_name]
    options, camera, bat, kernels = MG.import_reference()
    import util  # noqa: F401

    opt = MG.make_opt(options, case["yaml"], H=case["H"], W=case["W"],
                      n_voxel_init=case["over"]["train_schedule"]["n_voxel_init"], extra=case["over"])
    opt.output_path = "/tmp/jt_engine_trace_out"
    opt.visdom = False   # (opt.tb stays the yaml's dict: process_GT_images reads opt.tb.num_images; the writer is a stand-in)
    opt.resume = False
    opt.load = None
    opt.visualize_gradient = False
    opt.generate_video_iters = []
    for k in ("scalar", "vis", "val", "ckpt", "vis_pose", "vis_train"):
        if k in opt.freq:
            opt.freq[k] = 10 ** 9
    B = case["n_views"]
    seed = 5
    torch.manual_seed(seed)
    np.random.seed(seed)
    var = MG.make_var(opt, B, seed=seed, llff=case["llff"])
    # structured supervision instead of white noise (the 2-D blur cache and the Sobel masks see edges)
    g = torch.Generator().manual_seed(seed + 3)
    base = torch.rand(B, 3, case["H"] // 4, case["W"] // 4, generator=g)
    var.image = (torch.nn.functional.interpolate(base, size=(case["H"], case["W"]), mode="bilinear", align_corners=True)
                 + 0.1 * torch.rand(B, 3, case["H"], case["W"], generator=g)).clamp(0, 1).contiguous()
    var.pop("train_edge_masks")  # the loop makes its own (nerf.Model.get_edge_mask)

    m = bat.Model(opt)
    m.train_data = FakeTrainData(EasyDict(var))
    m.n_train_views = B
    m.build_networks(opt)
    with torch.no_grad():  # semi-transparent content instead of the near-empty initial field
        for p in m.graph.nerf.tensorf.density_plane:
            p.mul_(case["density_scale"])
    m.setup_optimizer(opt)
    m.restore_checkpoint(opt)
    m.setup_visualizer(opt)
    # the engine's side effects that are out of scope (validation renders, files, plots)
    m.validate = lambda *a, **k: None
    m.save_checkpoint = lambda *a, **k: None
    m.visualize_pose = lambda *a, **k: None
    m.visualize_train = lambda *a, **k: None
    import util_vis
    util_vis.tb_wandb_image = lambda *a, **k: None
    return case, opt, var, m, bat, camera, seed


def snapshot(case_name, stop_it, out_path):
    """A renderer fixture (tools/make_golden.py: run_case format) taken INSIDE the reference's loop: Model.train runs up to
    iteration `stop_it`, then that iteration's forward / loss / backward is recorded on the state the loop has reached (grid
    after its upsamplings, trained factors and poses, the schedule's near plane / blur / loss weights of that iteration).  The
    hand-built fixtures of make_golden.py never visited e.g. the iterations where the LLFF near plane crosses zero."""
    case, opt, var, m, bat, camera, seed = build_reference_model(case_name)

    class Stop(Exception):
        pass

    orig = m.train_iteration

    def until(opt_, var_, loader):
        if m.it == stop_it:
            raise Stop()
        return orig(opt_, var_, loader)

    m.train_iteration = until
    try:
        m.train(opt)
    except Stop:
        pass
    assert m.it == stop_it
    v = EasyDict(dict(m.train_data.all))
    v.image = m.blurred_gt_cached_images[1.0]
    v.train_edge_masks = m.blurred_edge_masks[1.0]
    tf = m.graph.nerf.tensorf
    extra = None
    if tf.alphaMask is not None:   # the loop has been through an alpha-mask update: the mask is part of the state
        extra = {"mask.alpha_volume": tf.alphaMask.alpha_volume[0, 0].numpy().copy(), "mask.aabb": tf.alphaMask.aabb.numpy().copy()}
    MG.run_case(bat, camera, opt, m.graph, v, "train", it=stop_it, progress=stop_it / opt.max_iter, out_path=out_path,
                llff=case["llff"], torch_seed=3, extra_arrays=extra)


def run(case_name, out_dir):
    case, opt, var, m, bat, camera, seed = build_reference_model(case_name)
    import model.tensorf_repr.tensorBase as tB
    import model.tensorf_repr.batBase as bB
    B = case["n_views"]

    init_state = {k: v.detach().clone() for k, v in m.graph.state_dict().items()}
    graph = m.graph
    nerf = graph.nerf

    rec = MG.Recorder()
    rec.install(tB, bB)
    its = []
    cur = {}

    # ---- recorders -------------------------------------------------------------------------------------------------
    orig_train_iteration = m.train_iteration

    def train_iteration_spy(opt_, var_, loader):
        tf = nerf.tensorf
        cur.clear()
        cur.update(it=int(m.it), n_rays=int(opt_.nerf.n_rays), progress=float(nerf.progress.data),
                   lr_groups=[float(gp["lr"]) for gp in m.optim.param_groups], lr_pose=float(m.optim_pose.param_groups[0]["lr"]),
                   pose_grad_accum_iter=int(opt_.optim.pose_grad_accum_iter) if hasattr(opt_.optim, "pose_grad_accum_iter") else 1,
                   grid=[int(x) for x in tf.gridSize.tolist()], n_samples=int(nerf.n_samples),
                   TV_density_weight=float(opt_.loss_weight.TV_density), TV_color_weight=float(opt_.loss_weight.TV_color),
                   graph_it=int(graph.it), pose_step=False, n_choice_before=len(rec.np_choice),
                   n_randint_before=len(rec.np_randint), n_coin_before=len(rec.coin), n_jitter_before=len(rec.rand_like),
                   image_is_scale=None, has_alpha_mask=tf.alphaMask is not None,
                   ray_sampling_strategy=str(opt_.nerf.ray_sampling_strategy))
        # which cached supervision the loop picked: the scale whose cached tensor IS var.image
        for sc, t in getattr(m, "blurred_gt_cached_images", {}).items():
            if t is var_.image:
                cur["image_is_scale"] = float(sc)
        # the L1 weight this iteration's summarize_loss applies, by the reference's own code on a unit loss
        probe = EasyDict(L1=torch.tensor(1.0), render=torch.tensor(0.0), TV_density=torch.tensor(0.0), TV_color=torch.tensor(0.0))
        cur["L1_weight"] = float(m.summarize_loss(opt_, None, probe).all)
        loss = orig_train_iteration(opt_, var_, loader)
        cur["loss"] = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in loss.items()}
        cur["se3_after"] = graph.se3_refine.weight.detach().clone()
        cur["lr_pose_after"] = float(m.optim_pose.param_groups[0]["lr"])
        cur["progress_after"] = float(nerf.progress.data)
        its.append(dict(cur))
        return loss

    m.train_iteration = train_iteration_spy

    orig_pose_step = m.optim_pose.step

    def pose_step_spy(*a, **k):
        cur["pose_step"] = True
        cur["lr_pose_at_step"] = float(m.optim_pose.param_groups[0]["lr"])
        cur["se3_grad_at_step"] = graph.se3_refine.weight.grad.detach().clone()
        return orig_pose_step(*a, **k)

    m.optim_pose.step = pose_step_spy

    def install_forward_spy():
        tf = nerf.tensorf
        orig = tf.forward

        def spy(opt_, **kw):
            cur["forward_kwargs"] = scalar_kwargs(kw)
            cur["near_far"] = [float(tf.near_far[0]), float(tf.near_far[1])]
            out = orig(opt_, **kw)
            cur["blur_active"] = tf.kernel_density is not None
            return out

        tf.forward = spy

    install_forward_spy()
    orig_update = nerf.update_schedule

    def update_spy(opt_, it):
        orig_update(opt_, it)
        its[-1]["after_update"] = dict(
            it_arg=int(it), grid=[int(x) for x in nerf.tensorf.gridSize.tolist()], n_samples=int(nerf.n_samples),
            lr_groups=[float(gp["lr"]) for gp in m.optim.param_groups], lr_basis=float(nerf.lr_basis),
            lr_index=float(nerf.lr_index), TV_density_weight=float(opt_.loss_weight.TV_density),
            TV_color_weight=float(opt_.loss_weight.TV_color), has_alpha_mask=nerf.tensorf.alphaMask is not None,
            aabb=[float(x) for x in nerf.tensorf.aabb.view(-1).tolist()],
            resolution_scale_init=[float(x) for x in opt_.train_schedule.resolution_scale_init])

    nerf.update_schedule = update_spy

    try:
        m.train(opt)
    finally:
        rec.uninstall()

    # ---- pack ------------------------------------------------------------------------------------------------------
    out = {}
    for k, v in init_state.items():
        out["init." + k] = v.cpu().numpy()
    for k, v in m.graph.state_dict().items():
        if k.startswith("se3_refine") or k.startswith("nerf.progress"):
            out["final." + k] = v.detach().cpu().numpy()
    out["in.pose_gt"] = var.pose.numpy()
    out["in.intr"] = var.intr.numpy()
    out["in.intr_inv"] = var.intr_inv.numpy()
    out["in.idx"] = var.idx.numpy()
    out["in.image"] = var.image.numpy()
    out["trace.se3_after"] = torch.stack([r.pop("se3_after") for r in its]).numpy()
    gsteps = [r.pop("se3_grad_at_step", None) for r in its]
    out["trace.se3_grad_at_step"] = torch.stack([g_ if g_ is not None else torch.zeros_like(gsteps[-1] if gsteps[-1] is not None else
                                                                                              graph.se3_refine.weight)
                                                 for g_ in gsteps]).numpy()
    jit = rec.rand_like
    for i, t in enumerate(jit):
        out["draw.jitter.%d" % i] = t.numpy()
    meta = dict(case=case_name, yaml=case["yaml"], n_views=B, H=case["H"], W=case["W"], llff=case["llff"],
                overrides=case["over"], density_scale=case["density_scale"], seed=seed,
                np_randint=rec.np_randint, np_choice=rec.np_choice, coin=rec.coin, n_jitter=len(jit),
                iterations=its, max_iter=int(opt.max_iter),
                final_grid=[int(x) for x in nerf.tensorf.gridSize.tolist()],
                reference="model/nerf.py:150-278 (Model.train) -> model/bat.py:96-116 -> model/base.py:154-172 -> "
                          "model/tensorf.py:399-447, run unmodified by tools/make_engine_trace.py")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(out_dir, "engine_trace_%s.npz" % case_name)
    np.savez_compressed(path, **out)
    print("wrote", path, "iterations", len(its), "final grid", meta["final_grid"],
          "pose steps", sum(1 for r in its if r["pose_step"]), "size %.0f KB" % (os.path.getsize(path) / 1024))
    for r in its[:3] + its[-2:]:
        print({k: r[k] for k in ("it", "n_rays", "progress", "lr_pose", "pose_step", "L1_weight", "loss")})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--case", default="all")
    ap.add_argument("--snapshot", default=None, help="CASE:ITERATION -> <out>/<case>_loop_it<ITERATION>.npz (a renderer fixture)")
    a = ap.parse_args()
    out_dir = os.path.abspath(a.out)
    os.makedirs(out_dir, exist_ok=True)
    if a.snapshot:
        cname, it = a.snapshot.split(":")
        snapshot(cname, int(it), os.path.join(out_dir, "%s_loop_it%d.npz" % (cname, int(it))))
        sys.exit(0)
    for name in (list(CASES) if a.case == "all" else [a.case]):
        if a.case == "all":  # one process per case: the reference keeps module-level state (opt edits, monkey patches)
            import subprocess
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--out", out_dir, "--case", name])
        else:
            run(name, out_dir)
    if a.case == "all":
        import subprocess
        for snap in SNAPSHOTS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--out", out_dir, "--snapshot", snap])
