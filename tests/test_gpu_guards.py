"""Non-finite guard (model/tensorf.py:43-44,147-151 without per-iteration host reads) and run-to-run reproducibility of
the float-atomic gradient scatters ("race detection")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(seed=0):
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views
    torch.manual_seed(seed)
    np.random.seed(seed)
    opt = make_options("bat_blender_VM", device=DEV, data=dict(image_size=[64, 64], num_views=4),
                       train_schedule=dict(n_voxel_init=24 ** 3, n_rays_init=256, n_rays_rest=256, upsample_iters=[10 ** 9]),
                       nerf=dict(n_rays=256), c2f_mode="None")
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=4)
    model.setup_optimizer(opt)
    with torch.no_grad():
        for p in model.graph.nerf.tensorf.density_plane:
            p.mul_(22.0)
    return opt, model, make_views(opt, 4, seed=3, device=DEV)


def test_nan_pose_and_infinite_loss_are_flagged_on_the_device():
    from joint_tensorf_amd import ops
    from joint_tensorf_amd.options import Opt
    opt, model, var = _model()
    ops.read_status(DEV)  # clear
    model.train_iteration(opt, Opt(dict(var)))
    model.check_finite(opt)  # clean iteration: nothing raised, no sync needed before
    with torch.no_grad():
        model.graph.se3_refine.weight[1, 2] = float("nan")
    model.train_iteration(opt, Opt(dict(var)))
    model.train_iteration(opt, Opt(dict(var)))  # the flag survives until somebody reads it
    with pytest.raises(FloatingPointError, match="camera pose"):
        model.check_finite(opt)
    model.check_finite(opt)  # cleared by the read
    opt2, model2, var2 = _model()
    bad = Opt(dict(var2))
    bad.image = var2.image.clone()
    bad.image[0, 0] = float("inf")   # one colour channel of a supervising view
    np.random.seed(5)
    model2.train_iteration(opt2, Opt(dict(bad)))   # (a second iteration would already carry NaN poses)
    with pytest.raises(FloatingPointError) as e:
        model2.check_finite(opt2)
    assert "loss" in str(e.value) and "camera pose" not in str(e.value)


def test_non_finite_pose_gradient_addends_cannot_pass_as_a_finite_gradient():
    """the ray (pose) gradients of the density walk are ALWAYS summed in 2^48 fixed point.  Round 3 replaced a
    non-finite addend by a poison magnitude inside the sum, which wraps (eight poisoned waves on one ray sum to 0 mod 2^64: a
    NaN gradient became exactly 0).  Now a bad addend is dropped and raises the library's sticky flag: while it is up every
    ray gradient jt_march_backward writes is NaN and the FINITE_GRAD bit goes into the bound status word."""
    from joint_tensorf_amd import ops
    from tests.test_gpu_edge import _batch
    from tests.test_gpu_fuzz import _scene
    tf, cfg, params, rs = _scene(41, "blender")
    o, d = _batch([("hit", 64)], seed=7)
    S = 512   # 16 runs of 32 samples per (ray, plane): more than eight poisoned waves per sum
    ops.read_status(DEV)

    def pose_grads(poison):
        og, dg = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
        out = tf(None, og, dg, white_bg=True, is_train=False, ndc_ray=False, N_samples=S)
        w = torch.ones_like(out[2])
        if poison:
            w[3] = float("nan")     # d loss / d opacity of ray 3 is NaN: every density-gradient addend of that ray's walk is
        (out[2] * w).sum().backward()  # (opacity only: the appearance path, whose float sums carry a NaN by themselves, is idle)
        return og.grad.clone(), dg.grad.clone()

    go, gd = pose_grads(False)
    assert torch.isfinite(go).all() and torch.isfinite(gd).all() and float(go.abs().max()) > 0
    assert ops.read_status(DEV) == 0
    go, gd = pose_grads(True)
    assert torch.isnan(go[3]).all() and torch.isnan(gd[3]).all(), "a poisoned ray came back finite: %s" % (go[3],)
    bits = ops.read_status(DEV)          # (clears the word and the sticky flag)
    assert bits & ops.FINITE_GRAD, bits
    go, gd = pose_grads(False)           # after the clear the same batch is clean again
    assert torch.isfinite(go).all() and torch.isfinite(gd).all()
    assert ops.read_status(DEV) == 0


def test_float_atomic_scatters_reproduce_within_rounding():
    """Two backward passes from the same state and draws: the factor / pose gradients come out of float atomics whose
    order differs from run to run; the difference must stay at summation-rounding level (a lost or doubled update --
    a race -- would show up at the size of one sample's contribution, 1e-3 or more of a texel's gradient)."""
    from joint_tensorf_amd.options import Opt
    opt, model, var = _model()
    tf = model.graph.nerf.tensorf
    jit = torch.rand(4096, 1, generator=torch.Generator().manual_seed(9)).to(DEV)
    runs = []
    for _ in range(3):
        np.random.seed(11)
        tf.jitter_override = jit
        g = model.graph
        g.it = model.it
        model.optim.zero_grad()
        model.optim_pose.zero_grad()
        v = g.forward(opt, Opt(dict(var)), mode="train")
        loss = model.summarize_loss(opt, v, g.compute_loss(opt, v, mode="train"))
        loss.all.backward()
        runs.append({k: p.grad.detach().clone() for k, p in g.named_parameters() if p.grad is not None})
    worst = ("", 0.0)
    for k in runs[0]:
        for other in runs[1:]:
            a, b = runs[0][k].double(), other[k].double()
            e = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
            if e > worst[1]:
                worst = (k, e)
            assert e <= 5e-6, (k, e)
    print("run-to-run gradient difference: worst %.2e of the tensor's max (%s)" % (worst[1], worst[0]))


def test_deterministic_mode_is_bit_reproducible():
    """JT_DETERMINISTIC (jt_set_deterministic): the scatters' float atomics become 64-bit fixed-point integer atomics
    (order-independent), the cross-block sums run in a fixed order -- three backward passes from the same state give
    BIT-IDENTICAL gradients and losses, and they agree with the default mode to rounding."""
    from joint_tensorf_amd._lib import lib
    from joint_tensorf_amd.options import Opt
    opt, model, var = _model()
    tf = model.graph.nerf.tensorf
    jit = torch.rand(4096, 1, generator=torch.Generator().manual_seed(9)).to(DEV)

    def backward():
        np.random.seed(11)
        tf.jitter_override = jit
        g = model.graph
        g.it = model.it
        model.optim.zero_grad()
        model.optim_pose.zero_grad()
        v = g.forward(opt, Opt(dict(var)), mode="train")
        loss = model.summarize_loss(opt, v, g.compute_loss(opt, v, mode="train"))
        loss.all.backward()
        out = {k: p.grad.detach().clone() for k, p in g.named_parameters() if p.grad is not None}
        out["loss.all"] = loss.all.detach().clone()
        out["loss.L1"] = loss.L1.detach().clone()
        return out

    plain = backward()
    prev = lib.jt_set_deterministic(1)
    try:
        runs = [backward() for _ in range(3)]
    finally:
        lib.jt_set_deterministic(prev)
    assert len(runs[0]) >= 22
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
    worst = ("", 0.0)
    for k in plain:
        a, b = plain[k].double(), runs[0][k].double()
        e = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
        if e > worst[1]:
            worst = (k, e)
        assert e <= 5e-6, (k, e)
    print("deterministic vs default mode: worst difference %.2e of the tensor's max (%s)" % (worst[1], worst[0]))
    # a full optimizer step in the mode: parameters identical between two models stepped from the same state
    states = []
    lib.jt_set_deterministic(1)
    try:
        for _ in range(2):
            o2, m2, v2 = _model()
            m2.graph.nerf.tensorf.jitter_override = jit
            for it in range(3):
                np.random.seed(20 + it)
                m2.train_iteration(o2, Opt(dict(v2)))
                m2.after_iteration(o2)
            states.append({k: p.detach().clone() for k, p in m2.graph.named_parameters()})
    finally:
        lib.jt_set_deterministic(prev)
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k
