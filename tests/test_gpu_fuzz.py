"""Randomised GPU parity: random non-cubic grids, sample counts, ray mixes (hits, grazes, misses), thresholds, blur
kernels, jitter and both MLP kinds -- the HIP path through BAT_VMSplit against the CPU oracle on identical state."""
import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O
from tests.test_gpu_edge import _batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(seed, kind):
    import joint_tensorf_amd as jt
    rs = np.random.RandomState(seed)
    grid = [int(rs.randint(9, 23)) for _ in range(3)]
    llff = kind == "llff"
    ca, app_dim, hid, mode = (20, 20, 32, "MLP_Fea_WeakView") if llff else (48, 27, 64, "MLP_Fea")
    act, shift = ("relu", 0.0) if llff else ("softplus", -10.0)
    aabb = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
    thres = float(10.0 ** rs.uniform(-7, -4))
    step_ratio = float(rs.uniform(0.3, 0.7))
    torch.manual_seed(seed)
    tf = jt.BAT_VMSplit(aabb, grid, DEV, density_n_comp=[16, 16, 16], appearance_n_comp=[ca] * 3, app_dim=app_dim,
                        near_far=[2.0, 6.0], shadingMode=mode, density_shift=shift, distance_scale=25.0, view_pe=2,
                        fea_pe=2, featureC=hid, step_ratio=step_ratio, fea2denseAct=act, rayMarch_weight_thres=thres,
                        volume_init_scale=0.1, volume_init_bias=0.0 if not llff else 0.2)
    with torch.no_grad():
        scale = float(rs.uniform(8, 30)) if not llff else float(rs.uniform(0.3, 1.5))
        for p in tf.density_plane:
            p.mul_(scale)
    sd = {k: v.detach().cpu().clone() for k, v in tf.state_dict().items()}
    params = O.params_from_state_dict(sd, prefix="")
    for _, v in O.flat_params(params):
        v.requires_grad_(True)
    cfg = O.SceneCfg(aabb, grid, [2.0, 6.0], step_ratio=step_ratio, density_shift=shift, distance_scale=25.0,
                     fea2denseAct=act, rayMarch_weight_thres=thres, shadingMode=mode, view_pe=2, fea_pe=2)
    return tf, cfg, params, rs


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("kind", ["blender", "llff"])
def test_random_scene_vs_oracle(seed, kind):
    tf, cfg, params, rs = _scene(100 * (kind == "llff") + seed, kind)
    S = int(rs.randint(17, 131))
    n_hit, n_graze, n_miss = int(rs.randint(1, 40)), int(rs.randint(0, 9)), int(rs.randint(0, 5))
    o, d = _batch([("hit", n_hit), ("graze", max(n_graze, 1)), ("miss", max(n_miss, 1))], seed=seed)
    R = o.shape[0]
    train = bool(rs.randint(2))
    blur = bool(rs.randint(2))
    white = bool(rs.randint(2))
    jit = torch.rand(R, 1, generator=torch.Generator().manual_seed(seed)) if train else None
    pd = pc = None
    kd = kc = None
    if blur:
        pd, pc = float(rs.uniform(0.02, 0.3)), float(rs.uniform(0.02, 0.3))
        kd, kc = O.get_kernel(cfg, pd, 16), O.get_kernel(cfg, pc, 16)
    oc, dc = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ref = O.render(cfg, params, oc, dc, S, white_bg=white, jitter=jit, kernel_density=kd, kernel_color=kc,
                   view_pe_progress=0.7, fea_pe_progress=0.4)
    og, dg = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    tf.jitter_override = jit.to(DEV) if train else None
    tf.coin_override = 0.9  # no random white background: `white` decides
    try:
        out = tf(None, og, dg, white_bg=white, is_train=train, ndc_ray=False, N_samples=S,
                 c2f_parameter_density=pd, c2f_parameter_color=pc, c2f_mode="uniform-gaussian" if blur else None,
                 c2f_kernel_size=16 if blur else None, view_pe_progress=0.7, fea_pe_progress=0.4)
    finally:
        tf.jitter_override = None
        tf.coin_override = None
    gc = torch.Generator().manual_seed(7 + seed)
    cot = [torch.randn(R, 3, generator=gc), torch.randn(R, generator=gc)]
    tot = (ref[0] * cot[0]).sum() + (ref[2] * cot[1]).sum()
    if tot.requires_grad:
        tot.backward()
    ((out[0] * cot[0].to(DEV)).sum() + (out[2] * cot[1].to(DEV)).sum()).backward()
    tag = "%s seed %d grid %s S %d R %d train %d blur %d" % (kind, seed, cfg.gridSize, S, R, train, blur)
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), ref[0].detach().numpy(), atol=3e-5, err_msg=tag)
    np.testing.assert_allclose(out[2].detach().cpu().numpy(), ref[2].detach().numpy(), atol=3e-5, err_msg=tag)
    np.testing.assert_allclose(out[1].detach().cpu().numpy(), ref[1].detach().numpy(), atol=2e-4, err_msg=tag)
    got = {}
    for grp in ("density_plane", "density_line", "app_plane", "app_line"):
        for i in range(3):
            got["%s.%d" % (grp, i)] = getattr(tf, grp)[i].grad
    got["basis_mat.weight"] = tf.basis_mat.weight.grad
    for k, t in zip(("w1", "b1", "w2", "b2", "w3", "b3"), tf.renderModule.weights()):
        got["mlp." + k] = t.grad
    for n, v in O.flat_params(params):
        r = torch.zeros_like(v) if v.grad is None else v.grad
        g = got[n]
        assert g is not None and torch.isfinite(g).all(), (tag, n)
        scale = max(float(r.abs().max()), 1e-30)
        assert float((g.cpu() - r).abs().max()) <= 3e-3 * scale + 1e-10, (tag, n, float((g.cpu() - r).abs().max()), scale)
    for a, b, n in ((og, oc, "g_o"), (dg, dc, "g_d")):
        r = torch.zeros_like(b) if b.grad is None else b.grad
        scale = max(float(r.abs().max()), 1e-30)
        assert float((a.grad.cpu() - r).abs().max()) <= 3e-3 * scale + 1e-10, (tag, n)


@pytest.mark.parametrize("seed", range(6))
def test_random_ndc_scene_vs_oracle(seed):
    """NDC rays (shared z row with one shared jitter row, |d| folded into the step length), LLFF-style box."""
    import joint_tensorf_amd as jt
    rs = np.random.RandomState(500 + seed)
    grid = [int(rs.randint(9, 20)) for _ in range(3)]
    aabb = [-1.5, -1.67, -2.0, 1.5, 1.67, 1.0]
    near_far = [float(rs.uniform(0.01, 0.4)), 1.0]
    thres = 1e-7
    torch.manual_seed(seed)
    tf = jt.BAT_VMSplit(aabb, grid, DEV, density_n_comp=[16, 16, 16], appearance_n_comp=[20] * 3, app_dim=20,
                        near_far=list(near_far), shadingMode="MLP_Fea_WeakView", density_shift=0.0, distance_scale=25.0,
                        view_pe=2, fea_pe=2, featureC=32, step_ratio=0.3, fea2denseAct="relu",
                        rayMarch_weight_thres=thres, volume_init_scale=0.05, volume_init_bias=0.2)
    with torch.no_grad():  # semi-transparent instead of opaque (an opaque field has denormal gradients)
        for pl in tf.density_plane:
            pl.mul_(float(rs.uniform(0.05, 0.3)))
    sd = {k: v.detach().cpu().clone() for k, v in tf.state_dict().items()}
    params = O.params_from_state_dict(sd, prefix="")
    for _, v in O.flat_params(params):
        v.requires_grad_(True)
    cfg = O.SceneCfg(aabb, grid, near_far, step_ratio=0.3, density_shift=0.0, distance_scale=25.0, fea2denseAct="relu",
                     rayMarch_weight_thres=thres, shadingMode="MLP_Fea_WeakView", view_pe=2, fea_pe=2)
    R, S = int(rs.randint(3, 50)), int(rs.randint(20, 120))
    g = torch.Generator().manual_seed(seed)
    o = torch.cat([2.6 * (torch.rand(R, 2, generator=g) - 0.5), -1.0 + 0.05 * torch.rand(R, 1, generator=g)], -1)
    d = torch.cat([1.2 * (torch.rand(R, 2, generator=g) - 0.5), 1.6 + 0.8 * torch.rand(R, 1, generator=g)], -1)
    train = bool(rs.randint(2))
    jit = torch.rand(1, S, generator=g) if train else None
    white = bool(rs.randint(2))
    oc, dc = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ref = O.render(cfg, params, oc, dc, S, white_bg=white, jitter=jit, ndc_ray=True)
    og, dg = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    tf.jitter_override = jit.to(DEV) if train else None
    tf.coin_override = 0.9
    try:
        out = tf(None, og, dg, white_bg=white, is_train=train, ndc_ray=True, N_samples=S)
    finally:
        tf.jitter_override = tf.coin_override = None
    cot = [torch.randn(R, 3, generator=g), torch.randn(R, generator=g)]
    ((ref[0] * cot[0]).sum() + (ref[2] * cot[1]).sum()).backward()
    ((out[0] * cot[0].to(DEV)).sum() + (out[2] * cot[1].to(DEV)).sum()).backward()
    tag = "ndc seed %d grid %s S %d R %d train %d" % (seed, grid, S, R, train)
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), ref[0].detach().numpy(), atol=3e-5, err_msg=tag)
    np.testing.assert_allclose(out[2].detach().cpu().numpy(), ref[2].detach().numpy(), atol=3e-5, err_msg=tag)
    np.testing.assert_allclose(out[1].detach().cpu().numpy(), ref[1].detach().numpy(), atol=2e-4, err_msg=tag)
    for a, b, n in ((og, oc, "g_o"), (dg, dc, "g_d")):
        scale = max(float(b.grad.abs().max()), 1e-30)
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 3e-3 * scale + 1e-10, (tag, n)
    for (n, v), p in zip(O.flat_params(params)[:3], tf.density_plane):
        scale = max(float(v.grad.abs().max()), 1e-30)
        assert float((p.grad.cpu() - v.grad).abs().max()) <= 3e-3 * scale + 1e-10, (tag, n)
