"""GPU unit parity of the individual C-ABI entry points against the CPU oracle / known answers."""
import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O
from tests.golden_util import GOLDEN, Fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_pose_known_answers():
    from joint_tensorf_amd import ops
    d = np.load(GOLDEN + "/known_answers.npz")
    wu = torch.tensor(d["se3.wu"], device=DEV, requires_grad=True)
    eye = torch.eye(3, 4, device=DEV)
    Rt = ops.train_pose(wu, None, eye)  # exp(se3) o identity
    np.testing.assert_allclose(Rt.detach().cpu().numpy(), d["se3.Rt"], atol=2e-6)
    (Rt * torch.tensor(d["se3.cot"], device=DEV)).sum().backward()
    np.testing.assert_allclose(wu.grad.cpu().numpy(), d["se3.grad_wu"], atol=5e-5, rtol=2e-4)


def test_pose_compose_vs_oracle():
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(3)
    B = 37
    se3 = (torch.randn(B, 6, generator=g) * 0.2)
    noise = O.se3_to_SE3(torch.randn(B, 6, generator=g) * 0.15)
    gt = O.se3_to_SE3(torch.randn(B, 6, generator=g))
    cot = torch.randn(B, 3, 4, generator=g)
    a = se3.clone().requires_grad_(True)
    ref = O.train_pose(a, noise, gt)
    (ref * cot).sum().backward()
    b = se3.clone().to(DEV).requires_grad_(True)
    out = ops.train_pose(b, noise.to(DEV), gt.to(DEV))
    (out * cot.to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-6)
    assert _rel(b.grad.cpu().numpy(), a.grad.numpy()) < 1e-4


@pytest.mark.parametrize("ndc", [False, True])
def test_raygen_vs_oracle(ndc):
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W = 5, 30, 40
    if ndc:
        pose = torch.eye(3, 4)[None].repeat(B, 1, 1) + 0.02 * torch.randn(B, 3, 4, generator=g)
    else:
        pose = O.se3_to_SE3(torch.randn(B, 6, generator=g))
        pose[..., 3] += torch.tensor([0.0, 0.0, 4.0])
    f = 0.8 * W
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]])[None].repeat(B, 1, 1)
    ray_idx = torch.randperm(H * W, generator=g)[:77]
    co, cd = torch.randn(B, 77, 3, generator=g), torch.randn(B, 77, 3, generator=g)
    p1 = pose.clone().requires_grad_(True)
    c, r = O.rays_for_pixels(p1, intr.inverse(), ray_idx, W)
    if ndc:
        c, r = O.convert_ndc(c, r, intr, near=1.0)
    ((c * co).sum() + (r * cd).sum()).backward()
    p2 = pose.clone().to(DEV).requires_grad_(True)
    c2, r2 = ops.ray_gen(p2, intr.inverse().to(DEV), intr.to(DEV), ray_idx.to(DEV), W, ndc=ndc, ndc_near=1.0)
    ((c2 * co.to(DEV)).sum() + (r2 * cd.to(DEV)).sum()).backward()
    np.testing.assert_allclose(c2.detach().cpu().numpy(), c.detach().numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(r2.detach().cpu().numpy(), r.detach().numpy(), atol=1e-5, rtol=1e-5)
    assert _rel(p2.grad.cpu().numpy(), p1.grad.numpy()) < 1e-4


@pytest.mark.parametrize("shape", [(16, 14, 14), (48, 21, 21), (4, 9, 9), (16, 70, 70)])
@pytest.mark.parametrize("sigma", [0.7, 2.3, 6.4])
def test_blur_plane_vs_oracle(shape, sigma):
    from joint_tensorf_amd import ops
    C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(1, C, H, W, generator=g)
    cot = torch.randn(1, C, H, W, generator=g)
    k = O.gaussian_kernel(sigma, 64)
    a = x.clone().requires_grad_(True)
    ref = O.blur_plane(k, a, W, H)  # cubic: (gm0, gm1) = (W, H)
    (ref * cot).sum().backward()
    b = ops.factor_logical(ops.factor_storage(x).to(DEV)).requires_grad_(True)
    out = ops.blur_factor(b, k.to(DEV))
    (out * cot.to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5, rtol=1e-5)
    assert _rel(b.grad.cpu().numpy(), a.grad.numpy()) < 1e-5


def test_blur_noncubic_reinterpret_vs_golden():
    """the reference's reshape quirk on a non-square plane (known answer captured from the reference)."""
    from joint_tensorf_amd import ops
    d = np.load(GOLDEN + "/known_answers.npz")
    x = torch.tensor(d["blur.noncubic.in"])  # [1,4,13,9] = [1,C,g[m1],g[m0]]
    k = torch.tensor(d["blur.kernel"])
    b = ops.factor_logical(ops.factor_storage(x).to(DEV))
    out = ops.blur_factor(b, k.to(DEV), True)
    assert tuple(out.shape) == tuple(d["blur.noncubic.out"].shape) == (1, 4, 9, 13)
    np.testing.assert_allclose(out.cpu().numpy(), d["blur.noncubic.out"], atol=2e-5, rtol=1e-5)
    # square plane: the re-interpretation is the identity (first 4 of the fixture's 5 channels; the kernels
    # take channel counts that are multiples of 4)
    cub = torch.tensor(d["blur.cubic.in"])[:, :4].contiguous()
    outc = ops.blur_factor(ops.factor_logical(ops.factor_storage(cub).to(DEV)), k.to(DEV), True)
    np.testing.assert_allclose(outc.cpu().numpy(), d["blur.cubic.out"][:, :4], atol=2e-5, rtol=1e-5)


def test_blur_line_vs_oracle():
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 16, 33, 1, generator=g)
    cot = torch.randn(1, 16, 33, 1, generator=g)
    k = O.gaussian_kernel(3.1, 64)
    a = x.clone().requires_grad_(True)
    ref = O.blur_line(k, a)
    (ref * cot).sum().backward()
    b = ops.factor_logical(ops.factor_storage(x).to(DEV)).requires_grad_(True)
    out = ops.blur_factor(b, k.to(DEV))
    (out * cot.to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5, rtol=1e-5)
    assert _rel(b.grad.cpu().numpy(), a.grad.numpy()) < 1e-5


@pytest.mark.parametrize("name", ["blender_train_mid", "blender_train_sharp", "llff_train_sharp"])
def test_march_stage_vs_oracle(name):
    """sigma_feat / weight / shade list of jt_march_forward against the oracle's intermediates."""
    from joint_tensorf_amd import ops
    from joint_tensorf_amd._lib import lib, ptr, check
    from tests.test_gpu_parity import build_scene
    fx = Fixture(name)
    m = fx.meta
    if m["c2f_mode"] is not None:
        pytest.skip("sharp fixtures only")
    cfg = fx.cfg()
    params = fx.params(requires_grad=False)
    center, ray = fx.t("mid.center"), fx.t("mid.ray_dir")
    jitter = fx.t("in.jitter") if m["is_train"] else None
    _, _, _, aux = O.render(cfg, params, center, ray, m["N_samples"], white_bg=fx.white_bg(), jitter=jitter,
                            ndc_ray=m["ndc_ray"], return_aux=True)
    tf = build_scene(fx, DEV, "torch")
    R, S = center.shape[0], m["N_samples"]
    g = m["gridSize"]
    rc = ops.RenderCfg(aabb=m["aabb"], plane_hw=[(g[ops.MAT_MODE[i][1]], g[ops.MAT_MODE[i][0]]) for i in range(3)],
                       line_len=[g[ops.VEC_MODE[i]] for i in range(3)], n_comp_density=m["density_n_comp"][0],
                       n_comp_app=m["app_n_comp"][0], step_size=m["stepSize"], near_far=m["near_far"],
                       distance_scale=m["distance_scale"], density_shift=m["density_shift"],
                       density_act=0 if m["fea2denseAct"] == "softplus" else 1,
                       weight_thres=m["rayMarch_weight_thres"], n_samples=S, ndc=m["ndc_ray"], white_bg=fx.white_bg(),
                       app_dim=m["app_dim"], mlp_kind=tf.renderModule.kind, mlp_hidden=m["featureC"],
                       view_pe=m["view_pe"], fea_pe=m["fea_pe"])
    scene = rc.scene()
    fac = ops._factors_struct([ops.factor_storage(p) for p in tf.density_plane],
                              [ops.factor_storage(p) for p in tf.density_line],
                              [ops.factor_storage(p) for p in tf.app_plane],
                              [ops.factor_storage(p) for p in tf.app_line])
    f32 = dict(device=DEV, dtype=torch.float32)
    o, d = center.to(DEV).contiguous(), ray.to(DEV).contiguous()
    zvals = jit = None
    if m["ndc_ray"]:
        zvals = torch.linspace(m["near_far"][0], m["near_far"][1], S)[None]
        if jitter is not None:
            zvals = zvals + jitter.view(1, -1) * ((m["near_far"][1] - m["near_far"][0]) / S)
        zvals = zvals.to(DEV).contiguous()
    elif jitter is not None:
        jit = jitter.to(DEV).contiguous().view(-1)
    sf, w, tmin = torch.empty(R, S, **f32), torch.empty(R, S, **f32), torch.empty(R, **f32)
    cnt = torch.empty(R, device=DEV, dtype=torch.int32)
    off = torch.empty(R + 1, device=DEV, dtype=torch.int32)
    sidx = torch.empty(R, S, device=DEV, dtype=torch.int16)
    op, dep = torch.empty(R, **f32), torch.empty(R, **f32)
    check(lib.jt_march_forward(scene, fac, ptr(o), ptr(d), ptr(jit), ptr(zvals), R, ptr(sf), ptr(w), ptr(tmin),
                               ptr(cnt), ptr(off), ptr(sidx), ptr(op), ptr(dep), None), "march")
    torch.cuda.synchronize()
    valid = aux["valid"]
    ref_feat = torch.zeros(R, S)
    ref_feat[valid] = aux["sigma_feat"]
    np.testing.assert_allclose(sf.cpu().numpy()[valid.numpy()], ref_feat.numpy()[valid.numpy()], atol=2e-5, rtol=2e-5)
    assert np.all(sf.cpu().numpy()[~valid.numpy()] == 0.0)
    np.testing.assert_allclose(w.cpu().numpy(), aux["weight"].numpy(), atol=2e-6, rtol=2e-5)
    ref_cnt = aux["app_mask"].sum(-1).to(torch.int32)
    assert torch.equal(cnt.cpu(), ref_cnt)
    assert int(off[R]) == int(ref_cnt.sum())
    sidx_np = sidx.cpu().numpy().view(np.uint16)
    for r_ in range(R):
        want = np.nonzero(aux["app_mask"][r_].numpy())[0]
        assert np.array_equal(sidx_np[r_, :len(want)], want)


@pytest.mark.parametrize("shape", [(16, 14, 14), (48, 23, 17), (16, 33, 1), (20, 45, 37), (16, 64, 5), (48, 32, 9)])  # H >= 32: the row-walking TV kernels
def test_factor_reg_vs_oracle(shape):
    """fused L1 / TV sums and their gradient against the oracle's torch ops."""
    from joint_tensorf_amd import ops
    C, H, W = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, C, H, W, generator=g)
    w = torch.tensor([0.3, 1.7, -0.4])
    a = x.clone().requires_grad_(True)
    ref = torch.stack([a.abs().sum(), ((a[:, :, 1:, :] - a[:, :, :-1, :]) ** 2).sum(),
                       ((a[:, :, :, 1:] - a[:, :, :, :-1]) ** 2).sum()])
    (ref * w).sum().backward()
    b = ops.factor_logical(ops.factor_storage(x).to(DEV)).requires_grad_(True)
    out = ops.factor_reg(b)
    (out * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5)
    assert _rel(b.grad.cpu().numpy(), a.grad.numpy()) < 1e-5
    # TV loss assembled from the sums equals the oracle's tv_loss
    tv_ref = O.tv_loss(x)
    s = out.detach().cpu()
    tv = 0.0
    if H > 1:
        tv = tv + s[1] / (C * (H - 1) * W)
    if W > 1:
        tv = tv + s[2] / (C * H * (W - 1))
    np.testing.assert_allclose(float(2 * tv), float(tv_ref), rtol=2e-5)


@pytest.mark.parametrize("masked", [False, True])
def test_render_loss_vs_oracle(masked):
    """fused GT gather + (edge-split) nanmean MSE, value and gradient."""
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(11)
    B, r, H, W = 5, 37, 20, 30
    rgb = torch.rand(B, r, 3, generator=g)
    image = torch.rand(B, 3, H, W, generator=g)
    ray_idx = torch.randperm(H * W, generator=g)[:r]
    mask = (torch.rand(B, H * W, generator=g) < 0.4).to(torch.uint8)
    a = rgb.clone().requires_grad_(True)
    img_at = image.view(B, 3, -1).permute(0, 2, 1)[:, ray_idx]
    ref = O.render_loss(a, img_at, mask[:, ray_idx] if masked else None, 1.5, 0.5)
    (ref * 0.7).backward()
    b = rgb.clone().to(DEV).requires_grad_(True)
    out = ops.render_loss(b, image.to(DEV), ray_idx.to(DEV), mask.to(DEV) if masked else None, 1.5, 0.5)
    (out * 0.7).backward()
    np.testing.assert_allclose(float(out.detach()), float(ref.detach()), rtol=2e-6)
    assert _rel(b.grad.cpu().numpy(), a.grad.numpy()) < 1e-5


def test_tv_depth_value_vs_oracle():
    """jt_tv_depth_forward (model/tensorf.py:126-135, value only) against the oracle's restatement, ragged lattice shapes"""
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, H, W in ((18, 15, 15), (3, 1, 7), (2, 9, 1), (100, 5, 4)):
        d = torch.rand(B, H * W, 1, generator=g) * 7.0
        got = float(ops.tv_depth_value(d.to(DEV), B, H, W))
        ref = float(O.tv_depth(d, B, H, W))
        assert abs(got - ref) <= 2e-6 * max(1.0, abs(ref)), (B, H, W, got, ref)


def test_reg_losses_vs_oracle():
    """one-call L1 / TV_density / TV_color and their gradients against the oracle's formulas."""
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(13)
    grid = [9, 11, 10]
    p = O.init_params(grid, density_n_comp=(16, 16, 16), app_n_comp=(20, 20, 20), app_dim=20, featureC=32,
                      shadingMode="MLP_Fea_WeakView", scale=0.3, bias=-0.1, generator=g)
    for grp in ("density_plane", "density_line", "app_plane", "app_line"):
        p[grp] = [t - 0.05 for t in p[grp]]  # both signs, so that sign(x) matters
    leaves = [t.clone().requires_grad_(True) for grp in ("density_plane", "density_line", "app_plane", "app_line")
              for t in p[grp]]
    pp = dict(density_plane=leaves[0:3], density_line=leaves[3:6], app_plane=leaves[6:9], app_line=leaves[9:12])
    w = torch.tensor([0.3, 1.7, 0.9])
    ref = torch.stack([O.density_L1(pp), O.tv_planes(pp["density_plane"]), O.tv_planes(pp["app_plane"])])
    (ref * w).sum().backward()
    dev = [ops.factor_logical(ops.factor_storage(t.detach()).to(DEV)).requires_grad_(True) for t in leaves]
    out = ops.reg_losses(dev[0:3], dev[3:6], dev[6:9], dev[9:12], True, True)
    (out * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=3e-5)
    for i in range(9):
        assert _rel(dev[i].grad.cpu().numpy(), leaves[i].grad.numpy()) < 2e-5, i
    assert all(dev[i].grad is None for i in range(9, 12))


@pytest.mark.parametrize("tv", [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize("dev_weights", [False, True])
def test_reg_losses_fused_equals_forward_plus_backward(tv, dev_weights):
    """jt_reg_losses_fused (value and gradient of L1 / TV_density / TV_color in ONE launch, round 5) against the two-launch form
    it replaces in a training step, on LLFF-shaped factors (non-square planes of 16 / 20 channels, TV row walk and the short
    general loop): the three sums to float-sum noise, every gradient to 2e-6 of the tensor's maximum (the same expression per
    element up to the compiler's choice of fused multiply-adds), with the
    upstream gradients as host floats and as three floats in device memory."""
    import ctypes
    from joint_tensorf_amd import ops
    from joint_tensorf_amd._lib import lib, ptr, check
    g = torch.Generator().manual_seed(21)
    grid = [40, 45, 38]
    p = O.init_params(grid, density_n_comp=(16, 16, 16), app_n_comp=(20, 20, 20), app_dim=20, featureC=32,
                      shadingMode="MLP_Fea_WeakView", scale=0.3, bias=-0.1, generator=g)
    st = [[ops.factor_storage(t.detach() - 0.05).to(DEV).contiguous() for t in p[grp]]
          for grp in ("density_plane", "density_line", "app_plane", "app_line")]
    fac = ops._factors_struct(*st)
    hw = []
    for i in range(3):
        H, W, _ = st[0][i].shape
        hw += [H, W, st[1][i].shape[0]]
    hw_arr = (ctypes.c_int32 * 9)(*hw)
    w3 = [0.37, 1.9 if tv[0] else 0.0, 0.6 if tv[1] else 0.0]   # (a term that is not evaluated has weight zero in the run)
    stream = ops._stream()
    scratch = ops._reg_scratch(st[0][0].device)
    # two launches
    ga = [[torch.full_like(t, 7.0) for t in grp] for grp in st]
    out_a = torch.empty(3, device=DEV)
    check(lib.jt_reg_losses_forward(fac, hw_arr, 16, 20, int(tv[0]), int(tv[1]), ptr(scratch), ptr(out_a), stream), "fwd")
    g3 = torch.tensor(w3, device=DEV)
    check(lib.jt_reg_losses_backward(fac, hw_arr, 16, 20, ptr(g3), int(tv[0]), int(tv[1]), ops._factors_struct(*ga), 0,
                                     ptr(torch.empty(36, device=DEV)), stream), "bwd")
    # one launch
    gb = [[torch.full_like(t, 7.0) for t in grp] for grp in st]
    out_b = torch.empty(3, device=DEV)
    w_host = None if dev_weights else (ctypes.c_float * 3)(*w3)
    w_dev = ptr(g3) if dev_weights else None
    check(lib.jt_reg_losses_fused(fac, hw_arr, 16, 20, int(tv[0]), int(tv[1]), w_host, w_dev, ops._factors_struct(*gb),
                                  ptr(scratch), ptr(out_b), stream), "fused")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out_b.cpu().numpy(), out_a.cpu().numpy(), rtol=2e-6)
    for grp in range(4):
        for i in range(3):
            touched = grp < 2 or (grp == 2 and tv[1])
            if touched:
                # (a density plane without TV: the backward's element expression is the L1 term alone in both forms)
                assert _rel(gb[grp][i].cpu().numpy(), ga[grp][i].cpu().numpy()) < 2e-6, (grp, i)
            else:
                assert torch.all(gb[grp][i] == 7.0) and torch.all(ga[grp][i] == 7.0)
    assert int((scratch.view(torch.int32) != 0).sum()) == 0


def test_reg_losses_scratch_is_left_zero_and_values_repeat():
    """The one-launch regulariser forward sums into a persistent per-device scratch that its last workgroup reads AND resets
    with returning atomics behind relaxed tickets (csrc/jt_reg.hip: no release fence, the sums are memory-side atomics).  A
    partial sum that landed behind the reset would stay in the scratch and poison every later value of the process
    (ADVICE r4): after many calls on a large factor set (thousands of workgroups per call) the scratch is exactly zero and
    every call returned bit-identical values in deterministic mode / values within float-sum noise otherwise."""
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(5)
    grid = [96, 107, 96]
    p = O.init_params(grid, density_n_comp=(16, 16, 16), app_n_comp=(20, 20, 20), app_dim=20, featureC=32,
                      shadingMode="MLP_Fea_WeakView", scale=0.3, bias=-0.1, generator=g)
    dev = [ops.factor_logical(ops.factor_storage(t.detach()).to(DEV))
           for grp in ("density_plane", "density_line", "app_plane", "app_line") for t in p[grp]]
    outs = []
    with torch.no_grad():
        for _ in range(200):
            outs.append(ops.reg_losses(dev[0:3], dev[3:6], dev[6:9], dev[9:12], True, True))
    torch.cuda.synchronize()
    scratch = ops._reg_scratch(dev[0].device)
    assert int((scratch.view(torch.int32) != 0).sum()) == 0, "residue in the regularisers' scratch"
    vals = torch.stack(outs).double().cpu()
    ref = vals[0]
    assert torch.all((vals - ref).abs() <= 2e-6 * ref.abs()), (vals.min(0).values, vals.max(0).values)


def test_upsample_volume_grid_vs_golden():
    """BAT_VMSplit.upsample_volume_grid (channel-last parameters) against the reference's up_sampling_VM."""
    import joint_tensorf_amd as jt
    d = np.load(GOLDEN + "/known_answers.npz")
    g0 = [6, 7, 5]
    tf = jt.BAT_VMSplit([-1.5] * 3 + [1.5] * 3, g0, DEV, density_n_comp=[4, 4, 4], appearance_n_comp=[4, 4, 4],
                        app_dim=27, near_far=[2.0, 6.0], shadingMode="MLP_Fea", featureC=64)
    with torch.no_grad():
        for i in range(3):
            for grp in ("density", "app"):
                getattr(tf, grp + "_plane")[i].copy_(torch.tensor(d["up.plane_in.%d" % i], device=DEV))
                getattr(tf, grp + "_line")[i].copy_(torch.tensor(d["up.line_in.%d" % i], device=DEV))
    tf.upsample_volume_grid(d["up.res_target"].tolist())
    assert tf.gridSize.tolist() == d["up.res_target"].tolist()
    for i in range(3):
        for grp in ("density", "app"):
            p, l = getattr(tf, grp + "_plane")[i], getattr(tf, grp + "_line")[i]
            np.testing.assert_allclose(p.detach().cpu().numpy(), d["up.plane_out.%d" % i], atol=2e-6)
            np.testing.assert_allclose(l.detach().cpu().numpy(), d["up.line_out.%d" % i], atol=2e-6)
            # still channel-last storage behind the logical [1,C,H,W] shape
            assert p.permute(0, 2, 3, 1).is_contiguous() and l.permute(0, 2, 3, 1).is_contiguous()


def test_alpha_mask_update_and_shrink_vs_golden():
    """BAT_VMSplit.getDenseAlpha / updateAlphaMask / shrink against the reference fixture."""
    from tests.test_gpu_parity import build_scene
    fx = Fixture("blender_train_alphamask")
    tf = build_scene(fx, DEV, "mfma")
    tf.alphaMask = None
    tf.alphaMask_thres = float(fx.arrays["mask.thres"])
    grid = fx.arrays["mask.grid"].tolist()
    alpha, _ = tf.getDenseAlpha(grid)
    np.testing.assert_allclose(alpha.cpu().numpy(), fx.arrays["mask.dense_alpha"], atol=1e-7, rtol=5e-5)
    new_aabb = tf.updateAlphaMask(tuple(grid))
    vol = tf.alphaMask.alpha_volume[0, 0].cpu().numpy()
    assert (vol != fx.arrays["mask.alpha_volume"]).mean() < 2e-3   # voxels within round-off of the threshold
    np.testing.assert_allclose(new_aabb.cpu().numpy(), fx.arrays["mask.new_aabb"], atol=1e-6)
    # masked points sample to zero through the mask as the reference's AlphaGridMask does
    pts = torch.rand(500, 3, device=DEV) * 3 - 1.5
    ref = O.alpha_mask_sample((torch.tensor(fx.arrays["mask.alpha_volume"]), torch.tensor(fx.arrays["mask.aabb"])), pts.cpu())
    tf.alphaMask = type(tf.alphaMask)(DEV, fx.t("mask.aabb", DEV), fx.t("mask.alpha_volume", DEV))
    np.testing.assert_allclose(tf.alphaMask.sample_alpha(pts).cpu().numpy(), ref.numpy(), atol=1e-6)
    a_masked = tf.compute_alpha(pts, 0.1).cpu()
    assert float(a_masked[ref <= 0].abs().max()) == 0.0 and float(a_masked[ref > 0].max()) > 0.0
    # shrink with the reference's box: cropped channel-last factors, snapped box, grid
    fs = Fixture("blender_train_shrunk")
    tf.shrink(fx.t("mask.new_aabb", DEV))
    assert tf.gridSize.tolist() == fs.meta["gridSize"]
    np.testing.assert_allclose(tf.aabb.view(-1).cpu().numpy(), np.array(fs.meta["aabb"], np.float32), atol=1e-6)
    assert abs(float(tf.stepSize) - fs.meta["stepSize"]) < 1e-7
    assert tf.aabb.device.type == "cpu"
    sd = tf.state_dict()
    for k in fs.arrays:
        if k.startswith("param.nerf.tensorf.") and ("plane" in k or "line" in k):
            got = sd[k[len("param.nerf.tensorf."):]]
            assert tuple(got.shape) == fs.arrays[k].shape, k
            np.testing.assert_array_equal(got.cpu().numpy(), fs.arrays[k], err_msg=k)
            assert got.permute(0, 2, 3, 1).is_contiguous(), k


def test_alpha_mask_checkpoint_roundtrip():
    from tests.test_gpu_parity import build_scene
    fx = Fixture("blender_train_alphamask")
    tf = build_scene(fx, DEV, "mfma")
    ck = tf.save_param_state()
    assert ck["alphaMask.mask"].dtype == np.uint8 and tuple(ck["alphaMask.shape"])[-3:] == (20, 20, 20)
    vol = tf.alphaMask.alpha_volume.clone()
    tf.alphaMask = None
    tf.load_param_state(ck)
    assert torch.equal(tf.alphaMask.alpha_volume, vol)
    assert torch.equal(tf.alphaMask.aabb.cpu(), fx.t("mask.aabb"))


def test_vmadam_matches_torch_adam():
    """optim.VMAdam (one HIP launch for all tensors) against torch.optim.Adam: channel-last factor parameters, odd
    sizes, two lr groups, lr changes between steps, a parameter without gradient, state-dict interchange."""
    from joint_tensorf_amd.optim import VMAdam
    from joint_tensorf_amd.tensorf_repr import _channel_last_param
    g = torch.Generator().manual_seed(0)
    shapes = [(1, 16, 9, 7), (1, 48, 5, 1), (27, 144), (64,), (3,), (5, 5)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    ref_p = [torch.nn.Parameter(b.clone()) for b in base]
    hip_p = [_channel_last_param(b.clone().to(DEV)) if b.dim() == 4 else torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    groups = lambda ps: [dict(params=ps[:2], lr=0.02), dict(params=ps[2:], lr=1e-3)]
    ref = torch.optim.Adam(groups(ref_p), betas=(0.9, 0.99))
    hip = VMAdam(groups(hip_p), betas=(0.9, 0.99))
    for step in range(6):
        for i, (a, b) in enumerate(zip(ref_p, hip_p)):
            if i == 5 and step % 2 == 0:
                a.grad = b.grad = None  # a parameter that sits a step out keeps its own step count
                continue
            gr = torch.randn(a.shape, generator=g) * (10.0 ** (step - 3))
            a.grad = gr.clone()
            b.grad = gr.clone().to(DEV)  # contiguous gradient for a channel-last parameter: re-laid out by step()
        ref.step()
        hip.step()
        for grp_r, grp_h in zip(ref.param_groups, hip.param_groups):
            grp_r["lr"] *= 0.97
            grp_h["lr"] *= 0.97
        for i, (a, b) in enumerate(zip(ref_p, hip_p)):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=2e-7,
                                       err_msg="step %d tensor %d %s" % (step, i, tuple(a.shape)))
    sd = hip.state_dict()
    ref2 = torch.optim.Adam(groups(ref_p), betas=(0.9, 0.99))
    ref2.load_state_dict({"state": {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                                    for k, v in sd["state"].items()}, "param_groups": sd["param_groups"]})
    assert float(ref2.state[ref_p[0]]["step"]) == 6.0
    m_ref = ref.state[ref_p[0]]["exp_avg"].numpy()
    np.testing.assert_allclose(ref2.state[ref_p[0]]["exp_avg"].numpy(), m_ref, rtol=2e-6, atol=1e-7 * np.abs(m_ref).max())


def test_vmadam_planned_step_is_the_general_step():
    """optim.VMAdam keeps, after a step through the general path, the launch's item array and what it checked to build it
    (`_make_plan`); the next step re-verifies those facts and launches (`_planned_step`).  Two optimizers over copies of the same
    tensors, one with the plan switched off, through a history that keeps and breaks the plan: gradients rewritten in place
    (kept), the learning rate changed (kept), a gradient in a new tensor (broken), a parameter sitting a step out (broken),
    moments replaced by load_state_dict (broken) -- parameters and moments bit-identical after every step."""
    from joint_tensorf_amd import optim as jopt
    from joint_tensorf_amd.tensorf_repr import _channel_last_param
    g = torch.Generator().manual_seed(5)
    shapes = [(1, 16, 9, 7), (1, 48, 5, 1), (27, 144), (64,), (3,)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    mk = lambda: [_channel_last_param(b.clone().to(DEV)) if b.dim() == 4 else torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    pa, pb = mk(), mk()
    groups = lambda ps: [dict(params=ps[:2], lr=0.02), dict(params=ps[2:], lr=1e-3)]
    a, b = jopt.VMAdam(groups(pa), betas=(0.9, 0.99)), jopt.VMAdam(groups(pb), betas=(0.9, 0.99))

    def general_step(o):
        keep, jopt.PLAN_STEPS = jopt.PLAN_STEPS, False
        try:
            o.step()
        finally:
            jopt.PLAN_STEPS = keep
    assert jopt.PLAN_STEPS
    for x, y in zip(pa, pb):   # gradients in the parameters' own memory order: no layout copy, a plan can be made
        x.grad, y.grad = torch.empty_like(x, memory_format=torch.preserve_format), torch.empty_like(y, memory_format=torch.preserve_format)
    hits = []
    for step in range(12):
        for i, (x, y) in enumerate(zip(pa, pb)):
            gr = (torch.randn(x.shape, generator=g) * (10.0 ** (step % 5 - 3))).to(DEV)
            if step == 5 and i == 2:     # a gradient that lives somewhere else from now on
                x.grad, y.grad = torch.empty_like(x.grad), torch.empty_like(y.grad)
            if step == 8 and i == 4:     # a parameter that sits this step out ...
                x.grad = y.grad = None
                continue
            if step == 9 and i == 4:     # ... and comes back
                x.grad, y.grad = torch.empty_like(x), torch.empty_like(y)
            x.grad.copy_(gr)
            y.grad.copy_(gr)
        if step == 10:
            a.load_state_dict(a.state_dict())
            b.load_state_dict(b.state_dict())
        if step in (3, 7):
            for o in (a, b):
                for grp in o.param_groups:
                    grp["lr"] *= 0.5
        before = getattr(a, "planned_steps", 0)
        a.step()
        general_step(b)
        hits.append(getattr(a, "planned_steps", 0) - before)
        for i, (x, y) in enumerate(zip(pa, pb)):
            assert torch.equal(x.detach(), y.detach()), "step %d tensor %d" % (step, i)
            if x.grad is not None:
                assert a.state[x]["step"] == b.state[y]["step"]
                for k in ("exp_avg", "exp_avg_sq"):
                    assert torch.equal(a.state[x][k], b.state[y][k]), "step %d tensor %d %s" % (step, i, k)
    #       0  1  2  3  4  5  6  7  8  9 10 11      (a broken plan is re-made by the general step that follows)
    assert hits == [0, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0, 1], hits
    assert not hasattr(b, "planned_steps")


def test_adam_entry_points_by_value_and_device_coefficients_agree():
    """jt_adam_step (coefficients as launch arguments) and jt_adam_step_dyn (coefficients poked into device memory
    with jt_poke, the variant a hipGraph replays) are the same update (to the last-bit rounding of the coefficient)."""
    import ctypes
    import math
    from joint_tensorf_amd import ops
    from joint_tensorf_amd._lib import JtAdamItem, check, lib, ptr
    torch.manual_seed(3)
    sizes = [4096 + 8, 1000, 64]
    base = [(torch.randn(n, device=DEV), torch.randn(n, device=DEV), torch.rand(n, device=DEV), torch.rand(n, device=DEV))
            for n in sizes]
    lrs, t, b1, b2, eps = [1e-2, 5e-4, 3e-3], 7, 0.9, 0.99, 1e-8
    out = []
    for dyn_path in (False, True):
        ts = [[x.clone() for x in item] for item in base]
        arr = (JtAdamItem * len(ts))()
        coefs = []
        for k, (p, g, m, v) in enumerate(ts):
            arr[k].p, arr[k].g, arr[k].m, arr[k].v, arr[k].n = ptr(p), ptr(g), ptr(m), ptr(v), p.numel()
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            arr[k].lr, arr[k].bias_correction1, arr[k].bias_correction2 = lrs[k], bc1, bc2
            # what jt_adam_step derives from its float arguments, computed the same way
            f = ctypes.c_float
            coefs += [f(f(lrs[k]).value / f(bc1).value).value, f(1.0 / math.sqrt(f(bc2).value)).value]
        st = ops._stream()
        if dyn_path:
            dyn = torch.zeros(2 * len(ts), device=DEV)
            ops.poke_floats(dyn, coefs)
            check(lib.jt_adam_step_dyn(arr, len(ts), b1, b2, eps, ptr(dyn), st), "jt_adam_step_dyn")
        else:
            check(lib.jt_adam_step(arr, len(ts), b1, b2, eps, st), "jt_adam_step")
        torch.cuda.synchronize()
        out.append(ts)
    for a, b in zip(*out):
        for x, y in zip(a, b):
            torch.testing.assert_close(y, x, rtol=2e-6, atol=1e-9)
    # jt_poke: words arrive where they are sent, nothing around them is touched
    buf = torch.full((16,), -1, device=DEV, dtype=torch.int32)
    ops.poke_words(buf, [7, 11, 13], offset=5)
    assert buf.tolist() == [-1] * 5 + [7, 11, 13] + [-1] * 8
    assert lib.jt_poke(None, None, 1, None) != 0 and lib.jt_poke(ptr(buf), (ctypes.c_uint32 * 1)(), 257, None) != 0


def test_library_import_order_does_not_split_the_hip_runtime():
    """A fresh process that imports joint_tensorf_amd BEFORE torch (as __graft_entry__.build() followed by smoke()
    does) must still launch kernels on torch's tensors: the binding imports torch first so that the process has ONE HIP
    runtime (torch ships its own libamdhip64; libjt_render.so alone would pull /opt/rocm's in and every launch through
    it would fail with hipErrorNoDevice)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import joint_tensorf_amd\n"
            "from joint_tensorf_amd import ops\n"
            "import torch\n"
            "x = torch.zeros(4, device='cuda', dtype=torch.int32)\n"
            "ops.poke_words(x, [5, 6, 7])\n"
            "assert x.tolist() == [5, 6, 7, 0], x.tolist()\n"
            "print('ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])


def test_lattice_indices_from_device_offsets():
    """jt_lattice_indices (the all_view_rand_grid lattice of model/nerf.py:660-667 with its two offsets in device memory) against
    the host form Graph._lattice_at builds: every offset of a small stride, both lattice shapes an axis can take."""
    from joint_tensorf_amd import ops
    H, W, step = 37, 53, 7
    off = torch.zeros(6, device=DEV, dtype=torch.int32)
    for ox in range(step):
        for oy in (0, 2, step - 1):
            nx, ny = len(range(ox, W, step)), len(range(oy, H, step))
            ops.poke_words(off, [ox, oy])
            got = ops.lattice_indices(off, step, nx, ny, W).cpu()
            xs, ys = torch.arange(ox, W, step), torch.arange(oy, H, step)
            want = (xs[None, :] + ys[:, None] * W).reshape(-1)
            assert got.dtype == torch.int64 and torch.equal(got, want), (ox, oy)


def test_backward_runs_on_the_calling_thread():
    """ops.backward (the training loop's loss.backward(), model/base.py:162): the autograd engine walks the graph ON THE CALLING
    THREAD -- every node of this package's backward is a Python function, a worker thread only adds two hand-overs per iteration --
    with the same gradients as torch's default (a per-device worker thread)."""
    import threading
    from joint_tensorf_amd import ops
    seen = {}

    def run(on_caller):
        keep = ops.BACKWARD_ON_CALLER
        ops.BACKWARD_ON_CALLER = on_caller
        try:
            se3 = torch.zeros(2, 6, device=DEV, requires_grad=True)
            base = torch.eye(3, 4, device=DEV).repeat(2, 1, 1)
            pose = ops.train_pose(se3, None, base)

            def note(_g):
                seen.setdefault(on_caller, threading.get_ident())   # (a hook that returns None leaves the gradient as it is)
            pose.register_hook(note)
            w = torch.arange(24, device=DEV, dtype=torch.float32).view(2, 3, 4)
            ops.backward((pose * w).sum())
            return se3.grad.clone()
        finally:
            ops.BACKWARD_ON_CALLER = keep
    g_caller, g_worker = run(True), run(False)
    assert seen[True] == threading.get_ident() and seen[False] != threading.get_ident()
    assert torch.equal(g_caller, g_worker)
    assert ops.BACKWARD_ON_CALLER   # the default
