"""Pin the CPU oracle to the reference: every golden fixture (captured from the reference itself by
tools/make_golden.py) must be reproduced by oracle/tensorf_oracle.py to fp32 round-off."""
import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O
from tests.golden_util import CASES, Fixture, GOLDEN, replay_oracle

# tolerances (fp32): values abs, gradients relative to the tensor's max-abs
TOL_VAL = 2e-6
# 5e-5 of the tensor max holds on the machine the fixtures were captured on (Intel, 8 threads); the same replay on
# the GPU box's host (AMD EPYC: other vector code paths of exp / softplus) lands at 5.6e-5 for one tensor whose
# gradient is 1e-10 in size -- fp32 round-off either way
TOL_GRAD_REL = 1e-4


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("use_taps", [False, True])
def test_fixture_replay(name, use_taps):
    fx = Fixture(name)
    out = replay_oracle(fx, use_taps=use_taps)
    # manual taps sum in a different order than grid_sample's kernel: allow a wider band there
    tol_g = 5e-4 if use_taps else TOL_GRAD_REL
    if "dense" in name:
        # acc == 1 (saturated): the render gradient is a cancellation residue ~1e-3 of the L1 term
        tol_g = 5e-3 if use_taps else 1e-3  # 5e-4 on the capture host, 6.2e-4 on the GPU box's EPYC host
    np.testing.assert_allclose(out["pose"].detach().numpy(), fx.arrays["mid.current_pose"], atol=1e-6)
    np.testing.assert_allclose(out["center"].detach().reshape(-1, 3).numpy(), fx.arrays["mid.center"], atol=2e-6)
    np.testing.assert_allclose(out["ray"].detach().reshape(-1, 3).numpy(), fx.arrays["mid.ray_dir"], atol=2e-6)
    if out["kd"] is not None:
        np.testing.assert_allclose(out["kd"].numpy(), fx.arrays["mid.kernel_density"], atol=1e-7)
        np.testing.assert_allclose(out["kc"].numpy(), fx.arrays["mid.kernel_color"], atol=1e-7)
    np.testing.assert_allclose(out["rgb"].detach().numpy(), fx.arrays["out.rgb"], atol=TOL_VAL)
    np.testing.assert_allclose(out["opacity"].detach().numpy(), fx.arrays["out.opacity"], atol=TOL_VAL)
    np.testing.assert_allclose(out["depth"].detach().numpy(), fx.arrays["out.depth"], atol=1e-5)
    for k, v in out["losses"].items():
        np.testing.assert_allclose(float(v.detach()), float(fx.arrays["loss." + k]), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(float(out["total"].detach()), float(fx.arrays["loss.all"]), rtol=2e-5)
    for n, g in out["grads"].items():
        key = fx.grad_key(n)
        ref = fx.arrays[key]
        assert g is not None, n
        assert _rel(g.numpy(), ref) < tol_g, (n, _rel(g.numpy(), ref))
    assert _rel(out["grad_se3"].numpy(), fx.arrays["grad.se3_refine.weight"]) < tol_g


def test_known_answers():
    d = np.load(GOLDEN + "/known_answers.npz")
    wu = torch.tensor(d["se3.wu"], requires_grad=True)
    Rt = O.se3_to_SE3(wu)
    np.testing.assert_allclose(Rt.detach().numpy(), d["se3.Rt"], atol=1e-6)
    (Rt * torch.tensor(d["se3.cot"])).sum().backward()
    np.testing.assert_allclose(wu.grad.numpy(), d["se3.grad_wu"], atol=2e-5, rtol=1e-4)
    ab = O.compose_pair(torch.tensor(d["compose.a"]), torch.tensor(d["compose.b"]))
    np.testing.assert_allclose(ab.numpy(), d["compose.ab"], atol=1e-6)
    for i, s in enumerate(d["gauss.sigma"]):
        np.testing.assert_allclose(O.gaussian_kernel(s, 64).numpy(), d["gauss.k65"][i], atol=1e-7)
        np.testing.assert_allclose(O.gaussian_kernel(s, 8).numpy(), d["gauss.k9"][i], atol=1e-7)
    k = torch.tensor(d["blur.kernel"])
    cub = O.blur_plane(k, torch.tensor(d["blur.cubic.in"]), 11, 11)
    np.testing.assert_allclose(cub.numpy(), d["blur.cubic.out"], atol=1e-5)
    non = O.blur_plane(k, torch.tensor(d["blur.noncubic.in"]), 9, 13)
    assert non.shape == d["blur.noncubic.out"].shape == (1, 4, 9, 13)
    np.testing.assert_allclose(non.numpy(), d["blur.noncubic.out"], atol=1e-5)
    ln = O.blur_line(k, torch.tensor(d["blur.line.in"]))
    np.testing.assert_allclose(ln.numpy(), d["blur.line.out"], atol=1e-5)


def test_lattice_and_schedule_helpers():
    # 2048 rays over 100 views of 400x400 -> step 90; 65536 -> step 16 -> 625/view
    idx, step, gh, gw = O.rand_grid_ray_idx(400, 400, 2048, 100, 5, 7)
    assert step == 90 and len(idx) == gh * gw and 16 <= len(idx) <= 25
    idx, step, gh, gw = O.rand_grid_ray_idx(400, 400, 65536, 100, 0, 0)
    assert step == 16 and len(idx) == 625
    # stage table of bat_blender_VM
    n_voxels = [262144, 1036216, 4095998, 16190876, 64000012]
    res = [O.find_resolution([-1.5] * 3 + [1.5] * 3, n)[0] for n in n_voxels]
    assert res == [64, 101, 159, 252, 400]
    S = [O.find_n_samples([r] * 3, 0.5, 1000) for r in res]
    assert S == [221, 349, 550, 872, 1000]
    assert abs(O.interp_schedule(0.05, [0.3, 0.15, 0.07] + [0.0] * 8) - 0.225) < 1e-9
    assert O.resolve_blur(0.9, [0.3, 0.15, 0.07] + [0.0] * 8, [0.3, 0.15, 0.07] + [0.0] * 8, "train", 1.0) == (None, None)


# ---- N1: evaluation path (camera pre-alignment, test-time pose optimisation, sliced eval render) -------------
def _eval_fx():
    return Fixture("blender_test_optim")


def _sim3_of(fx):
    return dict(t0=fx.t("sim3.t0"), t1=fx.t("sim3.t1"), s0=fx.t("sim3.s0"), s1=fx.t("sim3.s1"), R=fx.t("sim3.R"))


def test_prealign_cameras_golden():
    fx = _eval_fx()
    pose = O.compose_pair(fx.t("param.pose_noise"), fx.t("in.pose_gt"))
    pose = O.compose_pair(O.se3_to_SE3(fx.t("param.se3_refine.weight")), pose)
    assert torch.allclose(pose, fx.t("mid.pose_all"), atol=2e-6)
    # seven cameras at different heights (the three scene cameras are coplanar once centred: their alignment
    # rotation is decided by round-off and is only an input of the test-time optimisation below)
    aligned, sim3 = O.prealign_cameras(fx.t("align.pose"), fx.t("align.pose_gt"))
    for k in ("t0", "t1", "s0", "s1", "R"):
        assert torch.allclose(sim3[k], fx.t("align.sim3." + k), atol=2e-6), k
    assert torch.allclose(aligned, fx.t("align.pose_aligned"), atol=5e-6)
    eR, et = O.camera_alignment_error(aligned, fx.t("align.pose_gt"))
    assert torch.allclose(eR, fx.t("align.err.R"), atol=2e-5)
    assert torch.allclose(et, fx.t("align.err.t"), atol=5e-6)


def test_test_time_optim_and_eval_render_golden():
    """NOTE: calibrated on the host the fixture was captured on (this build container).  Test-time optimisation renders
    without jitter, so the first sample of every ray lies exactly on the box face and the last bit of the CPU's pose
    / ray arithmetic decides whether it counts (discreteness note): on the GPU box's EPYC host the
    very first loss of the trace differs by 1e-3 from the captured one.  The driver runs the CPU suite here."""
    fx = _eval_fx()
    m = fx.meta
    cfg = fx.cfg()
    params = fx.params(requires_grad=False)
    sim3 = _sim3_of(fx)
    call = m["forward_calls_optim"][0]
    se3, pose_refine_test, tr_se3, tr_loss = O.test_time_optim(
        cfg, params, sim3, fx.t("in.test_pose"), fx.t("in.test_image"), fx.t("in.intr_inv"), m["H"], m["W"],
        m["n_rays"], m["N_samples"], m["np_randint"], m["lr_pose"], m["lr_pose_test"], m["lr_pose_test_end"],
        m["test_iter"], white_bg=call["white_bg"], view_pe_progress=call["view_pe_progress"],
        fea_pe_progress=call["fea_pe_progress"])
    assert np.allclose(tr_loss, fx.arrays["trace.loss_render"], atol=2e-6)
    # Adam normalises the gradient: every step moves each coordinate by ~lr, so the trace is a sign test of
    # the pose gradient and a value test of the optimiser arithmetic
    assert torch.allclose(tr_se3, fx.t("trace.se3"), atol=2e-6)
    assert torch.allclose(se3, fx.t("out.se3_refine_test"), atol=2e-6)
    # the eval render uses the refinement from before the last step (reference quirk, see O.test_time_optim)
    pose = O.eval_pose(sim3, fx.t("in.test_pose"), pose_refine_test)
    rgb, depth, acc = O.render_by_slices(cfg, params, pose, fx.t("in.intr_inv"), m["H"], m["W"], m["n_rays"],
                                         m["N_samples"], white_bg=m["forward_call_eval"]["white_bg"])
    assert torch.allclose(rgb, fx.t("out.rgb"), atol=2e-6)
    assert torch.allclose(depth, fx.t("out.depth"), atol=2e-5)
    assert torch.allclose(acc, fx.t("out.opacity"), atol=2e-6)
    psnr = -10 * O.mse_nanmean(rgb.view(-1, m["H"], m["W"], 3).permute(0, 3, 1, 2), fx.t("in.test_image")).log10()
    assert abs(float(psnr) - float(fx.arrays["out.psnr"])) < 1e-4


# ---- N3: 2-D blur cache of the supervising images + Sobel edge masks -----------------------------------------
def test_gt_blur_and_edge_masks_golden():
    fx = Fixture("gt_blur_edge")
    m = fx.meta
    blurred = O.process_gt_images(fx.t("in.images"), m["it"] / m["max_iter"], m["blur_2d_c2f_schedule"], m["scales"],
                                  m["blur_2d_c2f_kernel_size"], m["blur_2d_mode"])
    for sc in m["scales"]:
        assert torch.allclose(blurred[sc], fx.t("blur.%g" % sc), atol=2e-6), sc
    masks = O.edge_masks(blurred, m["hard_edge_mask_mean_thresh"], m["soft_edge_mask"])
    for sc in m["scales"]:
        ref = fx.t("mask.%g" % sc)
        # a pixel whose Sobel magnitude sits within round-off of the threshold may flip
        assert (masks[sc] != ref).float().mean() < 1e-3, sc
    assert torch.equal(blurred[0.0], fx.t("in.images"))  # zero width: the images themselves


def test_upsample_vm_known_answer():
    """tensoRF.py:274-295 on a non-cubic grid."""
    d = np.load(GOLDEN + "/known_answers.npz")
    planes = [torch.tensor(d["up.plane_in.%d" % i]) for i in range(3)]
    lines = [torch.tensor(d["up.line_in.%d" % i]) for i in range(3)]
    up_p, up_l = O.upsample_vm(planes, lines, d["up.res_target"].tolist())
    for i in range(3):
        assert torch.allclose(up_p[i], torch.tensor(d["up.plane_out.%d" % i]), atol=1e-6)
        assert torch.allclose(up_l[i], torch.tensor(d["up.line_out.%d" % i]), atol=1e-6)


# ---- N4: alpha-mask volume and AABB shrink ---------------------------------------------------------------------
def test_alpha_mask_update_and_shrink_golden():
    fx = Fixture("blender_train_alphamask")
    cfg, params = fx.cfg(), fx.params(requires_grad=False)
    grid = fx.arrays["mask.grid"].tolist()
    alpha, _ = O.dense_alpha(cfg, params, grid)
    assert torch.allclose(alpha, fx.t("mask.dense_alpha"), atol=1e-7, rtol=2e-5)
    (vol, aabb), new_aabb = O.update_alpha_mask(cfg, params, grid, float(fx.arrays["mask.thres"]))
    assert (vol != fx.t("mask.alpha_volume")).float().mean() < 1e-3  # voxels within round-off of the threshold
    assert torch.equal(aabb, fx.t("mask.aabb"))
    assert torch.allclose(new_aabb, fx.t("mask.new_aabb"), atol=1e-6)
    # shrink: the cropped factors, box and grid of the next fixture
    fs = Fixture("blender_train_shrunk")
    cfg2, p2 = O.shrink(cfg, params, fx.t("mask.new_aabb"), grid)
    assert cfg2.gridSize == fs.meta["gridSize"]
    assert torch.allclose(cfg2.aabb.view(-1), torch.tensor(fs.meta["aabb"]), atol=1e-6)
    assert abs(float(cfg2.stepSize) - fs.meta["stepSize"]) < 1e-7
    ref = fs.params(requires_grad=False)
    for (n, a), (_, b) in zip(O.flat_params(p2), O.flat_params(ref)):
        assert a.shape == b.shape and torch.equal(a, b), n
