"""GPU parity of the evaluation path through the host mirror of the reference interface
(bat_hip.Model): Procrustes pre-alignment of the training cameras, test-time photometric pose optimisation
of a held-out view with the pose-only backward, and the sliced full-image render + PSNR, against the fixture
captured from the reference (tools/make_golden.py: eval_case) and the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture

pytestmark = pytest.mark.gpu


def _model(fx, device="cuda"):
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    m = fx.meta
    opt = make_options("bat_blender_VM", device=device, data=dict(image_size=[m["H"], m["W"]]),
                       train_schedule=dict(n_voxel_init=int(np.prod(m["gridSize"]))), nerf=dict(n_rays=m["n_rays"]),
                       optim=dict(test_iter=m["test_iter"]))
    torch.manual_seed(0)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=fx.arrays["in.pose_gt"].shape[0])
    g = model.graph
    sd = {k[len("param."):]: fx.t(k, device) for k in fx.arrays if k.startswith("param.")}
    tf_sd = {k[len("nerf.tensorf."):]: v for k, v in sd.items() if k.startswith("nerf.tensorf.")}
    g.nerf.tensorf.load_state_dict(tf_sd, strict=True)
    with torch.no_grad():
        g.se3_refine.weight.copy_(sd["se3_refine.weight"])
        g.pose_noise.copy_(sd["pose_noise"])
    assert list(g.nerf.tensorf.gridSize.tolist()) == m["gridSize"]
    assert g.nerf.n_samples == m["N_samples"]
    model.it = g.it = m["it"]
    g.nerf.set_progress(m["progress"])
    return opt, model


def test_prealign_cameras():
    fx = Fixture("blender_test_optim")
    opt, model = _model(fx)
    pose, pose_GT = model.get_all_training_poses(opt, fx.t("in.pose_gt"))
    np.testing.assert_allclose(pose.cpu().numpy(), fx.arrays["mid.pose_all"], atol=2e-6)
    # seven cameras at different heights (see tools/make_golden.py: the three scene cameras are a degenerate
    # alignment problem)
    p7, g7 = fx.t("align.pose", "cuda"), fx.t("align.pose_gt", "cuda")
    aligned, sim3 = model.prealign_cameras(opt, p7, g7)
    for k in ("t0", "t1", "s0", "s1", "R"):
        np.testing.assert_allclose(torch.as_tensor(sim3[k]).cpu().numpy(), fx.arrays["align.sim3." + k], atol=5e-6,
                                   err_msg=k)
    np.testing.assert_allclose(aligned.cpu().numpy(), fx.arrays["align.pose_aligned"], atol=1e-5)
    err = model.evaluate_camera_alignment(opt, aligned, g7)
    np.testing.assert_allclose(err.R.cpu().numpy(), fx.arrays["align.err.R"], atol=5e-5)
    np.testing.assert_allclose(err.t.cpu().numpy(), fx.arrays["align.err.t"], atol=1e-5)


@pytest.mark.parametrize("fused", [False, True])
def test_test_time_optim_and_eval_render(fused):
    """fused = True: the same reference trace through the single-launch kernel (opt.optim.test_fused, csrc/jt_fused.hip)"""
    from joint_tensorf_amd import ops
    from joint_tensorf_amd.options import Opt
    fx = Fixture("blender_test_optim")
    m = fx.meta
    opt, model = _model(fx)
    opt.optim.test_fused = fused
    g = model.graph
    g.sim3 = Opt(t0=fx.t("sim3.t0", "cuda"), t1=fx.t("sim3.t1", "cuda"), s0=fx.t("sim3.s0", "cuda"),
                 s1=fx.t("sim3.s1", "cuda"), R=fx.t("sim3.R", "cuda"))
    var = Opt(idx=torch.arange(1, device="cuda"), pose=fx.t("in.test_pose", "cuda"), intr=fx.t("in.intr", "cuda"),
              intr_inv=fx.t("in.intr_inv", "cuda"), image=fx.t("in.test_image", "cuda"))
    # replay the reference's host draws: lattice offsets (two per iteration) and the blur-scale choices
    ints, choices = list(m["np_randint"]), list(m["np_choice"])
    orig_randint, orig_choice = np.random.randint, np.random.choice
    trace_se3, trace_loss = [], []
    orig_compute = g.compute_loss

    def compute_spy(opt_, v, mode=None):
        out = orig_compute(opt_, v, mode=mode)
        if mode == "test-optim":
            trace_se3.append(v.se3_refine_test.detach().clone())
            trace_loss.append(float(out.render.detach()))
        return out

    np.random.randint = lambda *a, **k: ints.pop(0)
    np.random.choice = lambda *a, **k: choices.pop(0)
    g.compute_loss = compute_spy
    try:
        res = model.evaluate_view(opt, var)
    finally:
        np.random.randint, np.random.choice = orig_randint, orig_choice
        g.compute_loss = orig_compute
    v = res.var
    # scene parameters are trainable again and received no gradient (pose-only backward)
    assert all(p.requires_grad for p in g.nerf.tensorf.parameters())
    assert all(p.grad is None for p in g.nerf.tensorf.parameters())
    print("loss trace", trace_loss, fx.arrays["trace.loss_render"])
    print("se3 trace", torch.stack(trace_se3).cpu().numpy().round(6), fx.arrays["trace.se3"].round(6))
    # Without jitter (test time) the first sample of every ray lies exactly on the AABB face and the in-box test
    # is decided by the last bit of o + d z (discreteness note).  The training fixtures pin the ray
    # values to take that coin toss out; here the pose moves every iteration, so a few rays gain or lose their
    # first sample relative to the reference's CPU run: the first iteration (identical pose) must agree to
    # round-off, the later ones to the size of that effect.
    np.testing.assert_allclose(trace_loss[0], fx.arrays["trace.loss_render"][0], atol=2e-5)
    np.testing.assert_allclose(trace_loss, fx.arrays["trace.loss_render"], atol=6e-4)
    # Adam normalises the gradient: the first step moves every coordinate by exactly lr * sign(g) (1e-3)
    t = torch.stack(trace_se3).cpu().numpy()
    np.testing.assert_allclose(t[:2], fx.arrays["trace.se3"][:2], atol=2e-6)
    np.testing.assert_allclose(t, fx.arrays["trace.se3"], atol=3e-4)
    np.testing.assert_allclose(v.se3_refine_test.detach().cpu().numpy(), fx.arrays["out.se3_refine_test"], atol=3e-4)
    assert abs(res.psnr - float(fx.arrays["out.psnr"])) < 0.05
    assert res.rgb_map.shape == (1, 3, m["H"], m["W"]) and res.invdepth_map.shape == (1, 1, m["H"], m["W"])
    # the sliced eval render on its own, at the reference's refinement (the one from before its last Adam step,
    # see the NOTE in evaluate_test_time_photometric_optim): pixels agree to round-off except the rays whose
    # first sample flips in / out of the box (above); those move by the first sample's contribution
    var2 = Opt(idx=torch.arange(1, device="cuda"), pose=fx.t("in.test_pose", "cuda"), intr=fx.t("in.intr", "cuda"),
               intr_inv=fx.t("in.intr_inv", "cuda"), image=fx.t("in.test_image", "cuda"))
    with torch.no_grad():
        var2.pose_refine_test = ops.train_pose(fx.t("trace.se3", "cuda")[-1], None, torch.eye(3, 4, device="cuda"))
        var2 = g.forward(opt, var2, mode="eval")
    d = np.abs(var2.rgb.cpu().numpy() - fx.arrays["out.rgb"]).max(-1).reshape(-1)
    print("eval render: median %.2e, within 5e-5: %.3f, max %.2e" % (np.median(d), (d < 5e-5).mean(), d.max()))
    assert np.median(d) < 1e-5 and (d < 5e-5).mean() > 0.6 and d.max() < 0.05  # measured: 1e-7, 0.74
    psnr2 = -10 * float(g.MSE_loss(var2.rgb.view(-1, m["H"], m["W"], 3).permute(0, 3, 1, 2), var2.image).log10())
    assert abs(psnr2 - float(fx.arrays["out.psnr"])) < 0.01


def test_gt_blur_cache_and_edge_masks():
    """SURVEY 8(f) N3 through bat_hip.Model: the 201-tap 2-D blur of the supervising images on the HIP blur kernel
    and the Sobel edge masks against the reference fixture."""
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    fx = Fixture("gt_blur_edge")
    m = fx.meta
    opt = make_options("bat_blender_VM", device="cuda", data=dict(image_size=[m["H"], m["W"]]),
                       train_schedule=dict(n_voxel_init=14 ** 3))
    assert list(opt.blur_2d_c2f_schedule) == m["blur_2d_c2f_schedule"] and opt.blur_2d_c2f_kernel_size == 201
    model = bat_hip.Model(opt)
    model.it = m["it"]
    images = fx.t("in.images", "cuda")
    blurred = model.process_GT_images(opt, images)
    assert sorted(blurred.keys()) == m["scales"]
    for sc in m["scales"]:
        np.testing.assert_allclose(blurred[sc].cpu().numpy(), fx.arrays["blur.%g" % sc], atol=3e-6, err_msg=str(sc))
    masks = model.get_edge_mask(opt, blurred)
    for sc in m["scales"]:
        got, ref = masks[sc].cpu().numpy(), fx.arrays["mask.%g" % sc]
        assert got.dtype == np.uint8 and got.shape == ref.shape
        assert (got != ref).mean() < 2e-3, sc  # pixels within round-off of the threshold may flip
    # the per-iteration choice: caches are (re)built on multiples of 500 and re-used in between
    model.it = 1000
    np.random.seed(3)
    img, mask, sc = model.select_supervision(opt, images)
    assert sc in m["scales"] and img.shape == images.shape and mask.shape == (images.shape[0], m["H"] * m["W"])
    cache = model.blurred_gt_cached_images
    model.it = 1001
    model.select_supervision(opt, images)
    assert model.blurred_gt_cached_images is cache


def test_evaluate_full_runs_end_to_end():
    """camera alignment + per-view test-time optimisation + eval render + PSNR through Model.evaluate_full."""
    from joint_tensorf_amd.options import Opt
    fx = Fixture("blender_test_optim")
    opt, model = _model(fx)
    views = [Opt(idx=torch.arange(1, device="cuda"), pose=fx.t("in.test_pose", "cuda"), intr=fx.t("in.intr", "cuda"),
                 intr_inv=fx.t("in.intr_inv", "cuda"), image=fx.t("in.test_image", "cuda")) for _ in range(2)]
    np.random.seed(0)
    out = model.evaluate_full(opt, views, fx.t("in.pose_gt", "cuda"))
    assert out.R_error.shape == (3,) and out.t_error.shape == (3,)
    assert len(out.views) == 2 and len(out.psnr_per_view) == 2 and np.isfinite(out.psnr)
    assert 5.0 < out.psnr < 15.0  # random target image: ~9 dB
    assert model.graph.sim3 is not None


def test_batched_test_time_optim_reproduces_the_serial_trajectories():
    """V held-out views optimised in ONE iteration each (Model.evaluate_test_time_photometric_optim_batched: a
    [V, 6] parameter, every view on its own lattice draws, a photometric mean per view) against the reference's serial loop
    (model/bat.py:265-292) on the same views from the same random stream, in JT_DETERMINISTIC mode (order-independent sums):
    every view's se(3) vector, its refinement pose and its PSNR are the serial run's, and the host random stream ends where
    the serial run leaves it."""
    from joint_tensorf_amd._lib import lib
    from joint_tensorf_amd.options import Opt
    from joint_tensorf_amd.synthetic import make_views
    fx = Fixture("blender_test_optim")
    opt, model = _model(fx)
    opt.optim.test_iter = 6
    g = model.graph
    g.sim3 = Opt(t0=fx.t("sim3.t0", "cuda"), t1=fx.t("sim3.t1", "cuda"), s0=fx.t("sim3.s0", "cuda"),
                 s1=fx.t("sim3.s1", "cuda"), R=fx.t("sim3.R", "cuda"))
    tv = make_views(opt, 3, seed=21, device="cuda")
    views = [Opt(idx=torch.arange(1, device="cuda"), pose=tv.pose[i:i + 1], intr=tv.intr[i:i + 1], intr_inv=tv.intr_inv[i:i + 1],
                 image=tv.image[i:i + 1]) for i in range(3)]
    prev = lib.jt_set_deterministic(1)
    try:
        np.random.seed(77)
        serial = [model.evaluate_test_time_photometric_optim(opt, Opt(dict(v))) for v in views]
        end_serial = np.random.get_state()[1][:6].tolist(), np.random.get_state()[2]
        np.random.seed(77)
        batched = model.evaluate_test_time_photometric_optim_batched(opt, [Opt(dict(v)) for v in views])
        end_batched = np.random.get_state()[1][:6].tolist(), np.random.get_state()[2]
    finally:
        lib.jt_set_deterministic(prev)
    assert end_serial == end_batched
    assert len(batched) == 3 and all(b.test_optimised for b in batched)
    for s, b in zip(serial, batched):
        a, c = s.se3_refine_test.detach(), b.se3_refine_test.detach()
        assert float(a.abs().max()) > 1e-3        # the optimisation moved the pose
        assert torch.equal(a, c), (a, c)
        assert torch.equal(s.pose_refine_test.detach(), b.pose_refine_test.detach())
    # and through evaluate_full: opt.optim.test_batch renders the same views to the same PSNR as the serial evaluation
    np.random.seed(78)
    ref = model.evaluate_full(opt, [Opt(dict(v)) for v in views], fx.t("in.pose_gt", "cuda"))
    opt.optim.test_batch = 3
    np.random.seed(78)
    out = model.evaluate_full(opt, [Opt(dict(v)) for v in views], fx.t("in.pose_gt", "cuda"))
    np.testing.assert_allclose(out.psnr_per_view, ref.psnr_per_view, atol=2e-3)
