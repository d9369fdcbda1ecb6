"""GPU edge cases of the renderer against the CPU oracle on hand-made ray batches: rays that miss the box, ragged
per-ray sample counts, a single ray, batches whose size is not a multiple of the wave tiling, a field so thin
that no sample passes the shading threshold, and the pose-only backward."""
import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O
from tests.golden_util import Fixture
from tests.test_gpu_parity import build_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rays(kind, n, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g)
    o = 4.0 * o / o.norm(dim=-1, keepdim=True)
    if kind == "hit":
        tgt = 0.6 * (torch.rand(n, 3, generator=g) - 0.5)
    elif kind == "graze":  # through the corner region of the box: a handful of in-box samples
        tgt = torch.sign(torch.randn(n, 3, generator=g)) * 1.45 + 0.04 * torch.randn(n, 3, generator=g)
    else:  # miss: looking away from the scene
        tgt = 3.0 * o
    d = tgt - o
    return o, d / d.norm(dim=-1, keepdim=True)


def _batch(spec, seed=0):
    os_, ds_ = zip(*[_rays(k, n, seed + i) for i, (k, n) in enumerate(spec)])
    return torch.cat(os_), torch.cat(ds_)


def _run_both(fx, o, d, density_scale=None, white_bg=True, want="all"):
    m = fx.meta
    tf = build_scene(fx, DEV, "mfma")
    params = fx.params()
    if density_scale is not None:
        with torch.no_grad():
            for p in tf.density_plane:
                p.mul_(density_scale)
            for p in params["density_plane"]:
                p.mul_(density_scale)
    cfg = fx.cfg()
    oc, dc = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ref = O.render(cfg, params, oc, dc, m["N_samples"], white_bg=white_bg)
    og, dg = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    if want == "pose":
        for p in tf.parameters():
            p.requires_grad_(False)
    out = tf(None, og, dg, white_bg=white_bg, is_train=False, ndc_ray=False, N_samples=m["N_samples"])
    gc = torch.Generator().manual_seed(99)
    cot = [torch.randn(ref[0].shape, generator=gc), torch.randn(ref[2].shape, generator=gc)]
    tot = (ref[0] * cot[0]).sum().add((ref[2] * cot[1]).sum())
    if tot.requires_grad:  # with no sample in the box the reference's graph is empty: all gradients are zero
        tot.backward()
    (out[0] * cot[0].to(DEV)).sum().add((out[2] * cot[1].to(DEV)).sum()).backward()
    return tf, params, ref, out, (oc, dc), (og, dg)


def _check_values(ref, out):
    for a, b, name, tol in zip(out, ref, ("rgb", "depth", "opacity"), (2e-5, 1e-4, 2e-5)):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), atol=tol, err_msg=name)


def _check_grads(tf, params, tol=2e-3):
    got = {}
    for grp in ("density_plane", "density_line", "app_plane", "app_line"):
        for i in range(3):
            got["%s.%d" % (grp, i)] = getattr(tf, grp)[i].grad
    got["basis_mat.weight"] = tf.basis_mat.weight.grad
    for k, t in zip(("w1", "b1", "w2", "b2", "w3", "b3"), tf.renderModule.weights()):
        got["mlp." + k] = t.grad
    for n, v in O.flat_params(params):
        g = got[n]
        assert g is not None and torch.isfinite(g).all(), n
        ref = torch.zeros_like(v) if v.grad is None else v.grad
        scale = max(float(ref.abs().max()), 1e-30)
        err = float((g.cpu() - ref).abs().max())
        assert err <= tol * scale + 1e-12, (n, err, scale)


def _check_ray_grads(cpu, gpu, tol=2e-3):
    for a, b, name in zip(gpu, cpu, ("g_o", "g_d")):
        ref = torch.zeros_like(b) if b.grad is None else b.grad
        scale = max(float(ref.abs().max()), 1e-30)
        assert torch.isfinite(a.grad).all(), name
        assert float((a.grad.cpu() - ref).abs().max()) <= tol * scale + 1e-12, name


@pytest.mark.parametrize("spec", [
    [("miss", 5)],                              # nothing in the box at all
    [("hit", 1)],                               # a single ray
    [("hit", 3), ("miss", 2), ("graze", 2)],   # ragged sample counts, 7 rays (not a multiple of 4)
    [("graze", 9), ("miss", 1), ("hit", 23)],  # 33 rays: one more than a tile of 32 shaded entries per ... ray mix
], ids=["all-miss", "single-ray", "ragged-7", "ragged-33"])
def test_ragged_batches(spec):
    fx = Fixture("blender_train_mid")
    o, d = _batch(spec, seed=11)
    tf, params, ref, out, cpu, gpu = _run_both(fx, o, d)
    _check_values(ref, out)
    if spec == [("miss", 5)]:
        assert float(out[2].detach().abs().max()) == 0.0 and float((out[0].detach() - 1.0).abs().max()) == 0.0  # white bg
    _check_grads(tf, params)
    _check_ray_grads(cpu, gpu)


def test_no_sample_passes_the_shading_threshold():
    """A strongly negative density feature (softplus(x - 10) -> ~0) puts every weight below
    rayMarch_weight_thres: the appearance kernels see zero entries, the density path still has its gradient."""
    fx = Fixture("blender_train_mid")
    o, d = _batch([("hit", 6)], seed=5)
    tf, params, ref, out, cpu, gpu = _run_both(fx, o, d, density_scale=-50.0)
    from joint_tensorf_amd import ops  # noqa: F401
    _check_values(ref, out)
    _check_grads(tf, params)
    assert all(float(p.grad.abs().max()) == 0.0 for p in tf.app_plane)


def test_pose_only_backward_matches_full():
    fx = Fixture("blender_train_mid")
    o, d = _batch([("hit", 10), ("graze", 3)], seed=3)
    _, _, ref, out_full, cpu, gpu_full = _run_both(fx, o, d)
    tf, _, _, out_pose, _, gpu_pose = _run_both(fx, o, d, want="pose")
    assert all(p.grad is None for p in tf.parameters())
    # the two backwards follow two instantiations of the forward (full records / pose-only records) whose fused
    # multiply-adds the compiler is free to contract differently: equal to 1e-6 of the gradient's scale, not bit for bit
    for a, b in zip(gpu_pose, gpu_full):
        ga, gb = a.grad.cpu().numpy(), b.grad.cpu().numpy()
        np.testing.assert_allclose(ga, gb, rtol=1e-5, atol=1e-6 * float(np.abs(gb).max()))
    _check_ray_grads(cpu, gpu_pose)


def test_pose_only_backward_from_the_march_s_stored_density_derivatives():
    """A render of which only the rays want a gradient goes through jt_march_forward_pose / jt_march_backward_pose: the forward
    march leaves d(density feature) / d(coordinates) of every in-box sample and the backward reads them back (the default); with
    ops.POSE_MARCH_DERIVATIVES off the backward gathers the density taps a second time (rounds 4-5).  Same ray gradients, and
    both agree with the oracle's."""
    from joint_tensorf_amd import ops
    fx = Fixture("blender_train_mid")
    o, d = _batch([("hit", 10), ("graze", 3), ("miss", 2)], seed=3)
    seen = []
    orig = ops.check

    def spy(rc, what):
        seen.append(what)
        return orig(rc, what)
    grads = {}
    keep = ops.POSE_MARCH_DERIVATIVES
    ops.check = spy
    try:
        for stored in (True, False):
            ops.POSE_MARCH_DERIVATIVES = stored
            del seen[:]
            _, _, _, _, cpu, gpu = _run_both(fx, o, d, want="pose")
            assert ("jt_march_backward_pose" in seen) == stored and ("jt_march_forward_pose" in seen) == stored, seen
            _check_ray_grads(cpu, gpu)
            grads[stored] = [t.grad.cpu().numpy() for t in gpu]
    finally:
        ops.check = orig
        ops.POSE_MARCH_DERIVATIVES = keep
    for a, b in zip(grads[True], grads[False]):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6 * float(np.abs(b).max()))
