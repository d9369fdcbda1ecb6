"""Helpers of tests/test_gpu_fullsize.py (and tools/diag_fullsize.py): one training forward + loss + backward of a
BASELINE.json-sized configuration through the product path (bat_hip.Model -> C ABI), and the same iteration through the
parity-pinned oracle on the same device, sliced over the rays, in fp32 or fp64."""
import json
import os

import numpy as np
import torch

from oracle import tensorf_oracle as O  # checker only

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# achieved errors are appended here (one JSON line per case) when the directory exists: the evidence file
OUT = os.path.join(ROOT, "gpurun_out", "fullsize_parity.jsonl")


def rel_max(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rel_quantile(a, b, q=0.999):
    """q-quantile of |a - b| over the elements, relative to max |b|: the agreement of the bulk of a tensor, blind to a
    handful of outliers"""
    d = (a.double() - b.double()).abs().flatten()
    k = min(d.numel(), max(1, int(round(q * d.numel()))))
    return float(d.kthvalue(k).values / b.double().abs().max().clamp_min(1e-30))


def build(config, stage=-1, it=None, n_rays=None, n_voxel_final=None, density_scale=None, overrides=None, seed=0):
    import bench
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views
    torch.manual_seed(seed)
    np.random.seed(seed)
    opt = make_options(config, device=DEV, **(overrides or {}))
    if n_voxel_final:
        opt.train_schedule.n_voxel_final = n_voxel_final
    stage, it0 = bench.stage_setup(opt, stage)
    if it is not None:
        it0 = it
    opt.nerf.n_rays = (opt.train_schedule.n_rays_init if it0 < opt.train_schedule.change_n_rays_after_n_iters
                       else opt.train_schedule.n_rays_rest)
    if n_rays:
        opt.nerf.n_rays = n_rays
    B = int(opt.data.num_views)
    model = bench.build_model(opt, it0, B)
    tf = model.graph.nerf.tensorf
    with torch.no_grad():
        if density_scale:
            for p in tf.density_plane:
                p.mul_(density_scale)
        model.graph.se3_refine.weight.copy_(0.01 * torch.randn(B, 6, device=DEV))
    var = make_views(opt, B, seed=3, device=DEV)
    return opt, model, var, it0


def oracle_params(tf, dtype=torch.float32):
    sd = {k: v.detach().clone().contiguous().to(dtype) for k, v in tf.state_dict().items()}
    p = O.params_from_state_dict(sd, prefix="")
    for _, v in O.flat_params(p):
        v.requires_grad_(True)
    return p


def oracle_cfg(opt, tf, dtype=torch.float32):
    a = opt.arch
    cfg = O.SceneCfg(tf.aabb.view(-1).tolist(), tf.gridSize.tolist(), [float(tf.near_far[0]), float(tf.near_far[1])],
                     step_ratio=opt.nerf.step_ratio, density_shift=a.density_shift, distance_scale=a.distance_scale,
                     fea2denseAct=a.feature_to_density_activation,
                     rayMarch_weight_thres=a.tensorf.rayMarch_weight_thres, shadingMode=a.shading.model,
                     view_pe=a.shading.view_pe, fea_pe=a.shading.fea_pe,
                     ndc_near_plane=float(a.get("ndc_near_plane", 1.0))).to(DEV)
    if dtype != torch.float32:
        # the fp32 VALUES of the scene constants (box, step size) in a wider type: the same sample positions
        for k in ("aabb", "aabbSize", "invaabbSize", "units", "stepSize"):
            setattr(cfg, k, getattr(cfg, k).to(dtype))
    return cfg


def read_relu_masks(tf, n):
    """The ReLU signs the product path took for its n shaded samples (entry order = ray-major, ascending sample), read
    back from the record workspace the training forward left (layout: jt_shade_record_layout).  Two [n, hidden] bool
    tensors (layer 1, layer 2)."""
    import ctypes
    from joint_tensorf_amd import ops
    from joint_tensorf_amd._lib import lib
    lay = (ctypes.c_int32 * 4)()
    assert lib.jt_shade_record_layout(ctypes.byref(tf.last_render_cfg.scene()), lay) == 0
    rows, row0, hid, tile = (int(v) for v in lay)
    ws = ops._WS[(str(tf.density_plane[0].device), "shade")]
    ntile = (n + tile - 1) // tile
    rec = ws[:ntile * rows * tile * 4].view(torch.int32).view(ntile, rows, tile)
    out = []
    for layer in range(2):
        m = torch.zeros(ntile * tile, hid, dtype=torch.bool, device=ws.device)
        for h in range(2):
            word = rec[:, row0 + 2 * layer + h, :].reshape(-1)
            for mt in range(hid // 32):
                for r in range(16):
                    unit = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h
                    m[:, unit] = ((word >> (mt * 16 + r)) & 1).bool()
        out.append(m[:n])
    return out


def run_hip(opt, model, var_all, offsets=(3, 5), blur_scale=1.0, coin=0.3):
    """One training forward + loss + backward through the product path.  Returns outputs, gradients and the draws /
    rays the oracle needs to repeat the iteration."""
    from joint_tensorf_amd import ops
    from joint_tensorf_amd.options import Opt
    g = model.graph
    tf = g.nerf.tensorf
    B = len(var_all.idx)
    H, W = opt.H, opt.W
    ndc = bool(opt.camera.ndc)
    S = g.nerf.n_samples
    step = g.lattice_step(opt, B)
    ox, oy = offsets
    assert ox < step and oy < step
    n_lat = len(range(ox, W, step)) * len(range(oy, H, step))
    R = B * n_lat
    gen = torch.Generator().manual_seed(11)
    jit = torch.rand(1, S, generator=gen) if ndc else torch.rand(R, 1, generator=gen)
    tf.jitter_override = jit.to(DEV)
    tf.coin_override = coin
    ints, ch = [ox, oy], [blur_scale]
    orig_randint, orig_choice = np.random.randint, np.random.choice
    np.random.randint = lambda *a, **k: ints.pop(0)
    np.random.choice = lambda *a, **k: ch.pop(0)
    try:
        g.it = model.it
        model.optim.zero_grad()
        model.optim_pose.zero_grad()
        var = g.forward(opt, Opt(dict(var_all)), mode="train")
        loss = g.compute_loss(opt, var, mode="train")
        loss = model.summarize_loss(opt, var, loss)
        loss.all.backward()
    finally:
        np.random.randint, np.random.choice = orig_randint, orig_choice
        tf.jitter_override = None
        tf.coin_override = None
    torch.cuda.synchronize()
    assert var.rgb.shape[:2] == (B, n_lat)
    grads = {}
    for grp in ("density_plane", "density_line", "app_plane", "app_line"):
        for i in range(3):
            grads["%s.%d" % (grp, i)] = getattr(tf, grp)[i].grad.detach().clone()
    grads["basis_mat.weight"] = tf.basis_mat.weight.grad.detach().clone()
    for k, t in zip(("w1", "b1", "w2", "b2", "w3", "b3"), tf.renderModule.weights()):
        grads["mlp." + k] = t.grad.detach().clone()
    grads["se3"] = g.se3_refine.weight.grad.detach().clone()
    ndc_near = float(opt.arch.get("ndc_near_plane", 1.0))
    offset, sidx = tf.last_render_cfg.shade_lists
    cnt = (offset[1:] - offset[:-1]).long()
    sel = torch.arange(S, device=DEV)[None] < cnt[:, None]
    shade_mask = torch.zeros(R, S, dtype=torch.bool, device=DEV)
    shade_mask[sel.nonzero()[:, 0], (sidx.to(torch.int32) & 0xFFFF).long()[sel]] = True
    assert int(shade_mask.sum()) == int(offset[-1])
    relu = read_relu_masks(tf, int(offset[-1]))
    with torch.no_grad():  # the rays the product path marched (values only), to pin the oracle's
        c_hip, r_hip = ops.ray_gen(var.current_pose.detach(), var_all.intr_inv, var_all.intr, var.ray_idx, W, ndc=ndc,
                                   ndc_near=ndc_near)
    return dict(rgb=var.rgb.detach().reshape(R, 3), depth=var.depth.detach().reshape(R),
                opacity=var.opacity.detach().reshape(R), total=float(loss.all.detach()), grads=grads,
                ctx=dict(B=B, n_lat=n_lat, R=R, S=S, ndc=ndc, ndc_near=ndc_near, jit=jit, ray_idx=var.ray_idx, c=c_hip,
                         r=r_hip, shade_mask=shade_mask, shade_offset=offset.long(), relu=relu, blur_scale=blur_scale, coin=coin, image=var.image, kernel_density=tf.kernel_density))


def run_oracle(opt, model, var_all, ctx, dtype=torch.float32, slice_rays=4096, pin_mask=True, linear_dtype=None,
               pin_relu=True):
    """The same iteration through the oracle, sliced over the lattice pixels (gradients accumulate: the loss is a
    sum over rays with a global normalisation), rays pinned to the values the product path marched."""
    from joint_tensorf_amd.model.bat_hip import interp_schedule
    if linear_dtype is not None:
        O.LINEAR_DTYPE = linear_dtype
        try:
            return run_oracle(opt, model, var_all, ctx, dtype, slice_rays, pin_mask, None, pin_relu)
        finally:
            O.LINEAR_DTYPE = None
    g = model.graph
    tf = g.nerf.tensorf
    B, n_lat, R, S, ndc = ctx["B"], ctx["n_lat"], ctx["R"], ctx["S"], ctx["ndc"]
    H, W = opt.H, opt.W
    progress = g.nerf.progress_host
    cfg = oracle_cfg(opt, tf, dtype)
    params = oracle_params(tf, dtype)
    se3 = g.se3_refine.weight.detach().clone().to(dtype).requires_grad_(True)
    pd = O.interp_schedule(progress, opt.c2f_schedule_density) * ctx["blur_scale"]
    pc = O.interp_schedule(progress, opt.c2f_schedule_color)
    kd = kc = None
    if opt.c2f_mode != "None" and max(pd, pc) >= 0.001:
        kd = O.get_kernel(oracle_cfg(opt, tf), pd, opt.c2f_kernel_size).to(DEV)
        kc = O.get_kernel(oracle_cfg(opt, tf), pc, opt.c2f_kernel_size).to(DEV)
        np.testing.assert_allclose(ctx["kernel_density"].cpu().numpy(), kd.cpu().numpy(), rtol=1e-6, atol=1e-9)
        kd, kc = kd.to(dtype), kc.to(dtype)
    else:
        assert ctx["kernel_density"] is None
    vpe = interp_schedule(progress, opt.c2f_view_pe_schedule) if "c2f_view_pe_schedule" in opt else 1.0
    fpe = interp_schedule(progress, opt.c2f_fea_pe_schedule) if "c2f_fea_pe_schedule" in opt else 1.0
    white = bool(opt.nerf.setbg_opaque) or ctx["coin"] < 0.5
    image = ctx["image"].view(B, 3, H * W).to(dtype)
    lw = opt.loss_weight
    first = opt.train_schedule.update_alphamask_iters[0]
    w_l1 = float(lw.L1.rest if model.it > first else lw.L1.init)
    ref = dict(rgb=torch.empty(B, n_lat, 3, device=DEV, dtype=dtype), depth=torch.empty(B, n_lat, device=DEV, dtype=dtype),
               opacity=torch.empty(B, n_lat, device=DEV, dtype=dtype))
    per_view = max(1, slice_rays // B) if kd is None else n_lat
    total, shaded, in_box, flips, tie, flips_on = 0.0, 0, 0, 0, 0.0, 0
    relu_rep = {}
    noise = g.pose_noise.detach().to(dtype) if (opt.data.dataset == "blender" and opt.camera.noise) else None
    intr_inv, intr = var_all.intr_inv.to(dtype), var_all.intr.to(dtype)
    jit = ctx["jit"].to(DEV).to(dtype)
    for a in range(0, n_lat, per_view):
        b = min(a + per_view, n_lat)
        if opt.data.dataset == "blender":
            pose = O.train_pose(se3, noise, var_all.pose.to(dtype))
        else:
            pose = O.train_pose(se3, None, torch.eye(3, 4, device=DEV, dtype=dtype))
        idx = ctx["ray_idx"][a:b]
        c, r = O.rays_for_pixels(pose, intr_inv, idx, W)
        if ndc:
            c, r = O.convert_ndc(c, r, intr, near=ctx["ndc_near"])
        c = c + (ctx["c"][:, a:b].to(dtype) - c).detach()
        r = r + (ctx["r"][:, a:b].to(dtype) - r).detach()
        n = b - a
        # per-ray jitter rows of this slice: ray (view v, lattice point k) is row v * n_lat + k of the batch
        j = jit if ndc else jit.view(B, n_lat, 1)[:, a:b].reshape(B * n, 1)
        pin = None
        if pin_mask:
            pin = ctx["shade_mask"].view(B, n_lat, S)[:, a:b].reshape(B * n, S)
        relu = None
        if pin is not None and pin_relu:
            # entries (shaded samples, ray-major) of this slice's rays: ray (v, k) is row v * n_lat + k of the batch
            rays = (torch.arange(B, device=DEV)[:, None] * n_lat + torch.arange(a, b, device=DEV)[None]).reshape(-1)
            off = ctx["shade_offset"]
            cnt = off[rays + 1] - off[rays]
            tot = int(cnt.sum())
            first = torch.cumsum(cnt, 0) - cnt
            ent = torch.arange(tot, device=DEV) - torch.repeat_interleave(first, cnt) + torch.repeat_interleave(off[rays], cnt)
            relu = (ctx["relu"][0][ent], ctx["relu"][1][ent])
        rgb, depth, acc, aux = O.render(cfg, params, c.reshape(-1, 3), r.reshape(-1, 3), S, white_bg=white, jitter=j,
                                        ndc_ray=ndc, kernel_density=kd, kernel_color=kc, view_pe_progress=vpe,
                                        fea_pe_progress=fpe, return_aux=True, app_mask_override=pin,
                                        relu_masks_override=relu, relu_report=relu_rep)
        shaded += int(aux["own_app_mask"].sum())
        in_box += int(aux["valid"].sum())
        if pin is not None:
            # samples the two sides decided differently must be near-ties of `weight > thres`
            diff = aux["own_app_mask"] != pin
            flips += int(diff.sum())
            flips_on += int((diff & aux["own_app_mask"]).sum())   # the oracle shades, the product path did not
            if diff.any():
                w = aux["weight"].detach()[diff]
                tie = max(tie, float((w / cfg.rayMarch_weight_thres - 1).abs().max()))
        del aux
        rgb = rgb.view(B, n, 3)
        tgt = image[:, :, idx].permute(0, 2, 1)
        part = float(lw.render) * ((rgb - tgt) ** 2).sum() / (R * 3)
        part.backward()
        total += float(part.detach())
        ref["rgb"][:, a:b] = rgb.detach()
        ref["depth"][:, a:b] = depth.detach().view(B, n)
        ref["opacity"][:, a:b] = acc.detach().view(B, n)
    reg = w_l1 * O.density_L1(params)
    if float(lw.TV_density or 0) != 0.0:
        reg = reg + float(lw.TV_density) * O.tv_planes(params["density_plane"])
    if float(lw.TV_color or 0) != 0.0:
        reg = reg + float(lw.TV_color) * O.tv_planes(params["app_plane"])
    reg.backward()
    total += float(reg.detach())
    torch.cuda.synchronize()
    grads = {nme: v.grad for nme, v in O.flat_params(params)}
    grads["se3"] = se3.grad
    return dict(rgb=ref["rgb"].reshape(R, 3), depth=ref["depth"].reshape(R), opacity=ref["opacity"].reshape(R),
                total=total, grads=grads, shaded=shaded, in_box=in_box, blur=kd is not None, flips=flips, flips_oracle_on=flips_on,
                tie=tie, relu=relu_rep)


def compare(name, opt, model, hip, ref, truth=None, ref_b=None):
    """error report of the product path against the fp32 oracle `ref`; with `truth` (the fp64 oracle) also the errors
    of both fp32 computations against it; with `ref_b` (the fp32 oracle with its Linear layers accumulated in fp64)
    the sensitivity of the reference algorithm itself to the GEMM summation order (rep["order"])."""
    tf = model.graph.nerf.tensorf
    ctx = hip["ctx"]
    rep = dict(case=name, rays=ctx["R"], samples_per_ray=ctx["S"], grid=tf.gridSize.tolist(), blur=ref["blur"],
               shaded=ref["shaded"], in_box=ref["in_box"], shaded_hip=int(ctx["shade_mask"].sum()), mask_flips=ref["flips"],
               mask_flips_oracle_on=ref.get("flips_oracle_on", 0), relu_flips_oracle_on=ref["relu"].get("flips_oracle_on", 0),
               mask_flip_max_rel_distance_from_threshold=ref["tie"], relu_units=ref["relu"].get("units", 0),
               relu_flips=ref["relu"].get("flips", 0), relu_flip_max_abs_preactivation=ref["relu"].get("max_abs", 0.0), loss_hip=hip["total"], loss_oracle=ref["total"], values={},
               grads={})
    for k in ("rgb", "opacity", "depth"):
        rep["values"][k] = float((hip[k] - ref[k]).abs().max())
    for nme, v in ref["grads"].items():
        rep["grads"][nme] = [rel_max(hip["grads"][nme], v), rel_l2(hip["grads"][nme], v)]
        if truth is not None:
            t = truth["grads"][nme]
            rep["grads"][nme] += [rel_max(hip["grads"][nme], t), rel_l2(hip["grads"][nme], t), rel_max(v, t), rel_l2(v, t)]
    rep["bulk"] = {nme: rel_quantile(hip["grads"][nme], v) for nme, v in ref["grads"].items()}
    if ref_b is not None:
        rep["order"] = {nme: [rel_max(v, ref_b["grads"][nme]), rel_l2(v, ref_b["grads"][nme])]
                        for nme, v in ref["grads"].items()}
    if truth is not None:
        rep["values64"] = {k: [float((hip[k] - truth[k]).abs().max()), float((ref[k] - truth[k]).abs().max())]
                           for k in ("rgb", "opacity", "depth")}
    report(rep)
    return rep


def drop_workspaces():
    """the persistent record workspace is sized for the largest batch so far (120 GB after the 62 500-ray case)"""
    from joint_tensorf_amd import ops
    ops._WS.clear()
    torch.cuda.empty_cache()


def report(rep):
    print("\n[fullsize] %s: %d rays x %d samples (%s in box, %s shaded), grid %s, blur %s; loss hip %.8f oracle %.8f" % (
        rep["case"], rep["rays"], rep["samples_per_ray"], rep.get("in_box"), rep.get("shaded"), rep["grid"], rep["blur"],
        rep["loss_hip"], rep["loss_oracle"]))
    print("   shaded by the product path %s; decided differently %s (farthest from the threshold: %.1e relative)" % (
        rep.get("shaded_hip"), rep.get("mask_flips"), rep.get("mask_flip_max_rel_distance_from_threshold", 0.0)))
    print("   ReLU signs: %s units, %s decided differently by the oracle (largest |pre-activation| among them %.1e)" % (
        rep.get("relu_units"), rep.get("relu_flips"), rep.get("relu_flip_max_abs_preactivation", 0.0)))
    print("   direction of the disagreements (oracle ON, product path off / the other way): shading mask %s / %s, ReLU %s / %s"
          % (rep.get("mask_flips_oracle_on"), (rep.get("mask_flips") or 0) - (rep.get("mask_flips_oracle_on") or 0),
             rep.get("relu_flips_oracle_on"), (rep.get("relu_flips") or 0) - (rep.get("relu_flips_oracle_on") or 0)))
    print("   max abs error vs fp32 oracle: " + ", ".join("%s %.2e" % kv for kv in rep["values"].items()))
    if "values64" in rep:
        print("   max abs error vs fp64 oracle (hip | fp32 oracle): " +
              ", ".join("%s %.2e | %.2e" % (k, v[0], v[1]) for k, v in rep["values64"].items()))
    for k, v in rep.get("grads", {}).items():
        s = "   grad %-18s vs fp32 oracle: max-rel %.2e l2-rel %.2e" % (k, v[0], v[1])
        if "bulk" in rep:
            s += "  99.9%%-quantile %.2e" % rep["bulk"][k]
        if "order" in rep:
            s += "   | oracle's own GEMM-order sensitivity %.2e / %.2e" % tuple(rep["order"][k])
        if len(v) > 2:
            s += "   vs fp64: hip %.2e / %.2e   fp32 oracle %.2e / %.2e" % tuple(v[2:6])
        print(s)


def record(rep):
    """append the report to the evidence file (gpurun_out/fullsize_parity.jsonl) when that directory exists"""
    try:
        if os.path.isdir(os.path.dirname(OUT)):
            with open(OUT, "a") as f:
                f.write(json.dumps(rep) + "\n")
    except OSError:
        pass
