"""hipGraph replay of the training iteration (joint_tensorf_amd/graphed.py) against the eager loop: same initial
state, same host draws, pinned jitter -- parameters, pose refinements and losses must follow the same trajectory
(differences: the order of float atomics only), across lattice shapes, the edge-loss parity and a signature change.

Tolerances (calibrated with tools/graph_calib.py): with the jitter pinned two EAGER runs of this loop agree to 3e-7
in the losses and 1e-6 (norm) / 4e-6 (largest element) in every parameter tensor after 24 iterations (float atomics
order only), and the graph-replayed run sits at exactly the same level.  (With the jitter OFF the first sample of a
ray lies exactly on the box face and the last bit of the optimised pose decides whether it counts: two eager runs
then jump 1e-4 apart in the loss within a few iterations.)"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _build(seed=0, blur=False):
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views
    B, HW = 3, 42   # 42 px, stride 8: 5 or 6 lattice points per axis -> four lattice shapes
    opt = make_options("bat_blender_VM", device=DEV, data=dict(image_size=[HW, HW], num_views=B),
                       train_schedule=dict(n_voxel_init=14 ** 3, n_rays_init=96, n_rays_rest=96), nerf=dict(n_rays=96))
    # Jitter ON, but pinned: without jitter the first sample of every ray lies exactly on the box face and the
    # last bit of the (optimised) pose decides whether it is in -- a coin toss that makes two runs of the same loop
    # differ by 1e-4 in the loss after a few iterations (discreteness note).
    if not blur:
        opt.c2f_schedule_density = [0.0, 0.0]   # sharp stage; blur=True keeps the yaml's schedule (factor blur with its
        opt.c2f_schedule_color = [0.0, 0.0]     # random density scale: the taps reach the graph through static memory)
    torch.manual_seed(seed)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=B)
    model.setup_optimizer(opt)
    with torch.no_grad():
        for p in model.graph.nerf.tensorf.density_plane:
            p.mul_(22.0)
        model.graph.se3_refine.weight.copy_(0.01 * torch.randn(B, 6, device=DEV))
    var0 = make_views(opt, B, seed=3, device=DEV)
    gj = torch.Generator().manual_seed(17)
    model.graph.nerf.tensorf.jitter_override = torch.rand(256, 1, generator=gj).to(DEV)   # rows [:R] are used
    return opt, model, var0


def _run(use_graph, K, it0=0, blur=False, swap_images=False, windows=None):
    from joint_tensorf_amd.graphed import GraphedTrainStep
    from joint_tensorf_amd.options import Opt
    opt, model, var0 = _build(blur=blur)
    model.it = it0
    model.graph.nerf.set_progress(it0 / opt.max_iter)
    np.random.seed(5)
    stepper = GraphedTrainStep(model, min_repeats=0) if use_graph else None
    if windows is not None:  # (WINDOW, WINDOW_LEAD, REPROBE) of the per-stage launch-mode choice, shrunk to the test's length
        stepper.WINDOW, stepper.WINDOW_LEAD, stepper.REPROBE = windows
    losses = []
    orig_randint = np.random.randint
    alt_image = (0.25 + 0.5 * var0.image).contiguous()  # a second supervising image set (another 2-D blur scale)
    for k in range(K):
        var = Opt(dict(var0))
        if swap_images and k % 3 == 1:
            var.image = alt_image
        # the first two iterations are eager on both sides, the very first on the densest lattice (offsets 0, 0):
        # the persistent workspaces reach their final size there, as a training run's first iterations make them
        if k == 0:
            np.random.randint = lambda *a, **kw: 0
        try:
            loss = stepper.train_iteration(opt, var, force_eager=k < 2) if use_graph else model.train_iteration(opt, var)
        finally:
            np.random.randint = orig_randint
        losses.append([float(loss.all.detach()), float(loss.render.detach()), float(loss.L1.detach())])
        model.after_iteration(opt)
    sd = {k: v.detach().clone() for k, v in model.graph.state_dict().items()}
    if use_graph:
        stepper.stats["decisions"] = list(stepper.decisions)
    if not use_graph:
        return np.array(losses), sd, dict(early=getattr(model.optim, "early_steps", 0)), np.random.get_state()[1][:8].copy()
    return np.array(losses), sd, stepper.stats, np.random.get_state()[1][:8].copy()


@pytest.mark.parametrize("it0,blur", [(0, False), (9000, False), (0, True), (9000, True)])
def test_graph_replay_follows_the_eager_trajectory(it0, blur):
    K = 16
    l_e, sd_e, _, rs_e = _run(False, K, it0, blur)
    l_g, sd_g, stats, rs_g = _run(True, K, it0, blur)
    if blur:
        from joint_tensorf_amd.model.bat_hip import interp_schedule
        from joint_tensorf_amd.options import make_options
        o = make_options("bat_blender_VM", device="cpu")
        assert interp_schedule(it0 / o.max_iter, o.c2f_schedule_color) >= 0.001  # the blur really is on at it0
    assert stats["captured"] >= 2 and stats["replayed"] >= K - 4, stats
    assert (rs_e == rs_g).all()  # the host random stream is consumed identically
    np.testing.assert_allclose(l_g, l_e, rtol=2e-5, atol=1e-9)
    for k in sd_e:
        a, b = sd_e[k].float(), sd_g[k].float()
        assert float((a - b).norm()) <= 1e-4 * (float(a.norm()) + 1e-12), k
        assert float((a - b).abs().max()) <= 1e-3 * (float(a.abs().max()) + 1e-12), k


@pytest.mark.parametrize("it0", [0, 9000])
def test_one_replayed_step_equals_one_eager_step(it0):
    """The strict check: two identical models take the same three eager iterations, then one takes an eager step and
    the other the same step replayed from a freshly captured hipGraph.  Compared are the loss and Adam's FIRST
    MOMENTS after the step -- linear in the step's gradients, so float-atomics noise stays at 1e-6 of the norm (and a
    sample whose weight sits on rayMarch_weight_thres, shaded in one run and not in the other, at 2e-3: seen in one run
    of five; the tolerance is 1e-2), while
    the parameter update itself is almost a sign step this early (a weight whose gradient is noise moves by +-lr: 1-2 %
    of an MLP matrix's update norm was observed between two eager runs).  A stale lattice offset or a missed
    zero-fill shows as an O(1) difference of the moments; the Adam coefficients are pinned by the trajectory test
    above and by tests/test_gpu_units.py."""
    from joint_tensorf_amd.graphed import GraphedTrainStep
    from joint_tensorf_amd.options import Opt
    runs = []
    for use_graph in (False, True):
        opt, model, var0 = _build()
        model.it = it0
        model.graph.nerf.set_progress(it0 / opt.max_iter)
        np.random.seed(5)
        stepper = GraphedTrainStep(model, min_repeats=0)
        orig_randint = np.random.randint
        np.random.randint = lambda *a, **kw: 0      # one lattice shape throughout
        try:
            for _ in range(3):
                stepper.train_iteration(opt, Opt(dict(var0)), force_eager=True)
                model.after_iteration(opt)
            loss = stepper.train_iteration(opt, Opt(dict(var0)), force_eager=not use_graph)
            model.after_iteration(opt)
        finally:
            np.random.randint = orig_randint
        moments = {}
        for n, p in model.graph.named_parameters():
            st = model.optim.state.get(p) or model.optim_pose.state.get(p)
            if st:
                moments[n] = (st["exp_avg"].detach().clone().float(), st["exp_avg_sq"].detach().clone().float())
        runs.append((moments, float(loss.all.detach()), dict(stepper.stats)))
    (m_e, l_e, _), (m_g, l_g, st) = runs
    assert st["replayed"] == 1 and st["captured"] == 1, st
    assert abs(l_g - l_e) <= 1e-5 * abs(l_e)
    assert len(m_e) >= 20 and set(m_e) == set(m_g)
    for k in m_e:
        for a, b in zip(m_e[k], m_g[k]):
            assert float((a - b).norm()) <= 1e-2 * (float(a.norm()) + 1e-20), (k, float((a - b).norm()), float(a.norm()))


def test_graph_is_dropped_when_the_optimizer_is_rebuilt():
    from joint_tensorf_amd.graphed import GraphedTrainStep
    from joint_tensorf_amd.options import Opt
    opt, model, var0 = _build()
    opt.train_schedule.upsample_iters = [12, 10 ** 9]
    model.graph.nerf.upsample_list = opt.train_schedule.upsample_iters
    np.random.seed(1)
    stepper = GraphedTrainStep(model, min_repeats=0)
    for _ in range(30):
        stepper.train_iteration(opt, Opt(dict(var0)))
        model.after_iteration(opt)
    assert stepper.stats["replayed"] >= 20
    res = model.graph.nerf.resolution
    assert res[0] > 14  # the grid was upsampled in the middle and the run went on (graphs re-captured)
    for p in model.graph.parameters():
        assert torch.isfinite(p).all()


def test_graph_captured_eval_render_matches_the_sliced_render():
    """BASELINE.json configs[4]: the whole sliced render of a view as one hipGraph, replayed for other views."""
    from joint_tensorf_amd.options import Opt
    opt, model, var0 = _build()
    opt.nerf.eval_slice_rays = 500      # several slices per image (42 x 42 = 1764 pixels)
    opt.nerf.n_rays = 96
    g = model.graph
    g.eval()
    outs = {}
    for use_graph in (False, True):
        opt.nerf.eval_graph = use_graph
        res = []
        with torch.no_grad():
            for i in (0, 1, 2, 0):
                var = Opt({k: (v[i:i + 1] if torch.is_tensor(v) and v.shape[:1] == (3,) else v) for k, v in dict(var0).items()})
                var.idx = torch.arange(1, device=DEV)
                v = g.forward(opt, var, mode="vis_eval")
                res.append((v.rgb.clone(), v.depth.clone(), v.opacity.clone()))
        outs[use_graph] = res
    assert g.eval_graph is not None and g.eval_graph.entry is not None
    for a, b in zip(outs[False], outs[True]):
        for x, y in zip(a, b):
            assert x.shape == y.shape
            torch.testing.assert_close(y, x, rtol=0, atol=0)   # same kernels, no atomics in the forward: bit-equal
    assert not torch.equal(outs[True][0][0], outs[True][1][0])  # different views really differ


def test_graph_replayed_test_time_pose_optimisation_matches_eager():
    """model/bat.py:265-292 with every iteration replayed from a hipGraph.  Against the SAME loop run eagerly (same
    kernels, same Adam arithmetic; the pose path has no order-dependent float atomics) the se(3) trajectory agrees to
    round-off; against the reference-shaped loop (torch's fused Adam) only as far as the on-the-face coin toss of
    the first sample of a ray allows once the two poses differ in the last bit."""
    from joint_tensorf_amd.graphed import GraphedTestOptim
    from joint_tensorf_amd.options import Opt
    res = {}
    for kind in ("reference-shaped", "same-loop-eager", "replayed"):
        opt, model, var0 = _build()
        opt.optim.test_iter = 16
        opt.optim.test_graph = kind != "reference-shaped"
        if kind == "same-loop-eager":
            model._test_optim_graph = GraphedTestOptim(model)
            model._test_optim_graph.force_eager = True
        g = model.graph
        g.sim3 = Opt(t0=torch.zeros(3, device=DEV), t1=torch.zeros(3, device=DEV), s0=1.0, s1=1.0,
                     R=torch.eye(3, device=DEV))
        np.random.seed(11)
        out = []
        for i in (0, 1):   # two held-out views through the same graphs
            var = Opt({k: (v[i:i + 1].clone() if torch.is_tensor(v) and v.shape[:1] == (3,) else v) for k, v in dict(var0).items()})
            var.idx = torch.arange(1, device=DEV)
            with torch.no_grad():   # a wrong start pose for the optimisation to correct
                var.pose[0, :, 3] += torch.tensor([0.02, -0.01, 0.015], device=DEV)
            v = model.evaluate_test_time_photometric_optim(opt, var)
            out.append((v.se3_refine_test.detach().clone(), v.pose_refine_test.detach().clone()))
        res[kind] = out
        if kind == "replayed":
            st = model._test_optim_graph.stats
            assert st["replayed"] >= 16 and st["captured"] >= 1, st
    for (se3_r, pr_r), (se3_e, pr_e), (se3_g, pr_g) in zip(res["reference-shaped"], res["same-loop-eager"], res["replayed"]):
        assert float(se3_e.abs().max()) > 1e-4          # the optimisation moved
        torch.testing.assert_close(se3_g, se3_e, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(pr_g, pr_e, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(se3_g, se3_r, rtol=0, atol=0.25 * float(se3_r.abs().max()))


def _build_llff(it0):
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views
    B = 3
    opt = make_options("bat_llff_VM_MLP", device=DEV, data=dict(image_size=[30, 40], num_views=B),
                       train_schedule=dict(n_voxel_init=2200, n_rays_init=90, n_rays_rest=90,
                                           upsample_iters=[10 ** 9]), nerf=dict(n_rays=90))
    torch.manual_seed(0)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=B)
    model.setup_optimizer(opt)
    with torch.no_grad():
        model.graph.se3_refine.weight.copy_(0.01 * torch.randn(B, 6, device=DEV))
    var0 = make_views(opt, B, seed=3, device=DEV)
    model.it = it0
    model.graph.nerf.set_progress(it0 / opt.max_iter)
    S = model.graph.nerf.n_samples    # NDC: one jitter row shared by all rays (tensorBase.py:554-571)
    model.graph.nerf.tensorf.jitter_override = torch.rand(1, S, generator=torch.Generator().manual_seed(17)).to(DEV)
    return opt, model, var0


@pytest.mark.parametrize("it0,switch", [(30000, 0), (7000, 0), (7000, 11)])
def test_graph_replay_of_llff_iterations_is_bit_identical_to_eager(it0, switch):
    """bat_llff_VM_MLP: it0 = 30 000, past the point where its near-plane schedule has settled (progress 0.5), and -- round
    3 -- it0 = 7 000, in the FIRST half of the run, which round 2 refused to capture: the near plane moves every iteration
    (the un-jittered depth row and the jitter scale reach the graph through static memory, GraphedTrainStep._zvals_static),
    the factor blur is on, and pose gradients accumulate over 8 iterations before a pose step (model/bat.py:103-106: two
    pose steps fall into the 20 iterations).  NDC rays, WeakView MLP,
    the white-background COIN of every training call (two graph variants, the draw handed to whichever path runs), TV
    weights that decay every iteration (read from device memory inside the graph), pose-lr warm-up bookkeeping.
    switch = 11: the accumulation period drops from 8 to 1 at iteration it0 + 11, which is not a multiple of
    8 -- three iterations' partial pose-gradient sum is pending at the switch and must enter the first single step on both paths.
    Run in JT_DETERMINISTIC mode, where neither path has order-dependent sums: every loss term of every iteration and
    every parameter after 20 iterations must be EXACTLY equal between the eager loop and the hipGraph-replayed one
    (in the default mode the two drift apart like two eager runs do: Adam turns atomics-order noise in a near-zero
    gradient into a full +-lr step)."""
    from joint_tensorf_amd._lib import lib
    from joint_tensorf_amd.graphed import GraphedTrainStep
    from joint_tensorf_amd.options import Opt
    K = 20
    res = []
    prev = lib.jt_set_deterministic(1)
    try:
        for use_graph in (False, True):
            opt, model, var0 = _build_llff(it0)
            if switch:
                opt.train_schedule.change_n_AccumPoseGrad_after_n_iters = it0 + switch
            np.random.seed(5)
            torch.manual_seed(123)   # the coin stream
            stepper = GraphedTrainStep(model, min_repeats=0) if use_graph else None
            losses, tvw, depths = [], [], []
            for k in range(K):
                model.before_iteration(opt)
                var = Opt(dict(var0))
                loss = stepper.train_iteration(opt, var, force_eager=k < 2) if use_graph else model.train_iteration(opt, var)
                losses.append([float(loss[t].detach()) for t in ("all", "render", "L1", "TV_density", "TV_color")])
                depths.append((stepper.last_var if use_graph else var).depth.detach().flatten().clone())
                model.after_iteration(opt)
                tvw.append(float(opt.loss_weight.TV_density))
            sd = {k_: v.detach().clone() for k_, v in model.graph.state_dict().items()}
            res.append((np.array(losses), sd, stepper.stats if use_graph else None, float(torch.rand((1,))), tvw,
                        None if stepper is None else (stepper._lw.cpu().tolist(), list(model.fused_loss_weights(opt))),
                        depths))
    finally:
        lib.jt_set_deterministic(prev)
    (l_e, sd_e, _, r_e, tv_e, _, d_e), (l_g, sd_g, stats, r_g, tv_g, lw, d_g) = res
    # the depth map carries "- near_far[0] + 0.05" (batBase.py:147-150): under replay the near plane of THIS iteration, not the
    # one the launch was captured with (JtScene.near_plane_dev)
    for k in range(K):
        assert torch.equal(d_e[k], d_g[k]), "depth map of iteration %d" % k
    assert stats["replayed"] >= K - 8 and stats["captured"] >= 2, stats
    assert r_e == r_g and tv_e == tv_g and tv_e[0] > tv_e[-1] > 0   # same coin stream consumed, TV weights decaying
    if it0 < 20000:  # the first half of the run: pose steps every 8th iteration, the near plane on its way down
        assert int(opt.optim.pose_grad_accum_iter) == (1 if switch else 8) and 0.0 < float(opt.nerf.depth.range[0]) < 0.4
    np.testing.assert_array_equal(l_g, l_e)
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k
    # the device-side loss weights of the last replay are the host schedule's of that iteration (one decay step behind now)
    dec = model.graph.nerf.lr_decay_factor
    np.testing.assert_allclose(lw[0][2], lw[1][2] / dec, rtol=1e-6)


def test_supervising_image_set_changes_without_a_new_capture():
    """The reference supervises with one of five blur scales of the images per iteration (model/nerf.py:209-227): the
    buffer's address reaches the replayed graph through device memory, so alternating image sets neither re-capture nor
    change the trajectory (deterministic mode: exact equality with the eager loop)."""
    from joint_tensorf_amd._lib import lib
    prev = lib.jt_set_deterministic(1)
    try:
        l_e, sd_e, _, _ = _run(False, 14, 9000, False, swap_images=True)
        l_g, sd_g, stats, _ = _run(True, 14, 9000, False, swap_images=True)
        l_s, _, stats_same, _ = _run(True, 14, 9000, False, swap_images=False)
    finally:
        lib.jt_set_deterministic(prev)
    assert stats["captured"] == stats_same["captured"], (stats, stats_same)   # no graph per image set
    assert stats["replayed"] >= 8, stats
    assert not np.array_equal(l_g, l_s)                                      # the other image set did supervise
    np.testing.assert_array_equal(l_g, l_e)
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k


@pytest.mark.parametrize("it0,blur", [(9000, False), (0, True)])
def test_graph_replay_is_bit_identical_to_eager_in_deterministic_mode(it0, blur):
    """The Blender trajectory of test_graph_replay_follows_the_eager_trajectory in JT_DETERMINISTIC mode: exact equality."""
    from joint_tensorf_amd._lib import lib
    prev = lib.jt_set_deterministic(1)
    try:
        l_e, sd_e, _, rs_e = _run(False, 14, it0, blur)
        l_g, sd_g, stats, rs_g = _run(True, 14, it0, blur)
    finally:
        lib.jt_set_deterministic(prev)
    assert stats["replayed"] >= 8, stats
    assert (rs_e == rs_g).all()
    np.testing.assert_array_equal(l_g, l_e)
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k


def test_launch_mode_choice_times_both_paths_and_leaves_the_trajectory_alone():
    """GraphedTrainStep's per-stage choice between replay and eager launch (timed windows of each, graphed.py): with the
    windows shrunk to three iterations the 40 iterations of this run pass through replay windows, eager windows, decisions
    and re-probes -- and, in JT_DETERMINISTIC mode, end bit-identical to the plain eager loop with the same host random
    stream, whatever was chosen."""
    from joint_tensorf_amd._lib import lib
    prev = lib.jt_set_deterministic(1)
    try:
        l_e, sd_e, _, rs_e = _run(False, 40, 9000, False)
        l_g, sd_g, stats, rs_g = _run(True, 40, 9000, False, windows=(3, 1, 5))
    finally:
        lib.jt_set_deterministic(prev)
    assert len(stats["decisions"]) >= 2, stats
    assert all(c in ("eager", "replay") and te > 0 and tr > 0 for _, _, c, te, tr in stats["decisions"]), stats
    assert stats["eager_by_choice"] >= 2 * 4 and stats["replayed"] >= 2 * 4, stats   # both kinds of window ran at least twice
    assert (rs_e == rs_g).all()
    np.testing.assert_array_equal(l_g, l_e)
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k


@pytest.mark.parametrize("it0", [0, 9000])
def test_early_optimizer_step_of_the_appearance_factors_leaves_the_trajectory_alone(it0):
    """ops.ADAM_EARLY: an eager iteration steps the appearance factors on the auxiliary stream as soon as the appearance backward
    is through, beside the density backward (optim.VMAdam.step), everything else behind it as before.  Same trajectory as with
    the one-launch step on the launch stream, within float-atomics noise (the tolerances of the replay test above), and every
    iteration did take the early step."""
    from joint_tensorf_amd import ops
    K = 16
    assert ops.ADAM_EARLY
    l_a, sd_a, st_a, rs_a = _run(False, K, it0)
    ops.ADAM_EARLY = False
    try:
        l_b, sd_b, st_b, rs_b = _run(False, K, it0)
    finally:
        ops.ADAM_EARLY = True
    assert st_a["early"] >= K - 1 and st_b["early"] == 0, (st_a, st_b)   # (the step that creates the moments stays on one stream)
    assert (rs_a == rs_b).all()
    np.testing.assert_allclose(l_a, l_b, rtol=2e-5, atol=1e-9)
    for k in sd_a:
        a, b = sd_a[k].float(), sd_b[k].float()
        assert float((a - b).norm()) <= 1e-4 * (float(a.norm()) + 1e-12), k
        assert float((a - b).abs().max()) <= 1e-3 * (float(a.abs().max()) + 1e-12), k
