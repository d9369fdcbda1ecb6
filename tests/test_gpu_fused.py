"""The single-launch render + loss + backward-to-the-rays kernel of test-time pose optimisation (csrc/jt_fused.hip, row X1
of the coverage table) against (1) the STAGED HIP path it replaces -- march / shade / composite / loss kernels + autograd
with the pose-only backward, itself pinned to the reference's golden vectors (tests/test_gpu_parity.py, test_gpu_eval.py) --
and (2) the oracle's stock torch ops, on the same frozen scene, views, lattice and refinement: loss, colours, depth, opacity
and d loss / d se3.  Blender (MLP_Fea, softplus, white background) and LLFF (NDC rays, WeakView MLP, relu, non-cubic grid);
two views at once; run-to-run bit-reproducibility of the gradient (plain stores, no atomics)."""
import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(config, B, hw, grid, n_rays, dens_scale):
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views
    opt = make_options(config, device=DEV, data=dict(image_size=list(hw), num_views=B),
                       train_schedule=dict(n_voxel_init=grid, n_rays_init=n_rays, n_rays_rest=n_rays), nerf=dict(n_rays=n_rays))
    torch.manual_seed(0)
    np.random.seed(0)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=B)
    g = model.graph
    with torch.no_grad():
        for p in g.nerf.tensorf.density_plane:
            p.mul_(dens_scale)
    g.nerf.set_progress(1.0)          # after training: blur off, LLFF near plane settled
    var = make_views(opt, B, seed=5, device=DEV)
    eye = torch.eye(3, device=DEV)
    g.sim3 = __import__("joint_tensorf_amd.options", fromlist=["Opt"]).Opt(
        t0=torch.zeros(3, device=DEV), t1=torch.zeros(3, device=DEV), s0=torch.tensor(1.0, device=DEV),
        s1=torch.tensor(1.0, device=DEV), R=eye)
    return opt, model, var


def _iteration(opt, model, var, se3, fused, seed):
    """one test-optim forward + backward (model/bat.py:283-287); returns loss.render, outputs, d loss.all / d se3"""
    from joint_tensorf_amd import ops
    from joint_tensorf_amd.options import Opt
    g = model.graph
    opt.optim.test_fused = fused
    np.random.seed(seed)
    se3 = se3.detach().clone().requires_grad_(True)
    v = Opt(dict(var))
    v.se3_refine_test = se3
    eye = torch.eye(3, 4, device=DEV)
    frozen = [p for p in g.parameters() if p.requires_grad]
    for p in frozen:
        p.requires_grad_(False)
    try:
        v.pose_refine_test = ops.train_pose(se3, None, eye)
        v = g.forward(opt, v, mode="test-optim")
        assert (v.get("fused_render_loss") is not None) == fused
        loss = g.compute_loss(opt, v, mode="test-optim")
        loss = model.summarize_loss(opt, v, loss)
        loss.all.backward()
    finally:
        for p in frozen:
            p.requires_grad_(True)
    return dict(render=float(loss.render.detach()), rgb=v.rgb.detach().clone(), depth=v.depth.detach().clone(),
                opacity=v.opacity.detach().clone(), g=se3.grad.detach().clone(), ray_idx=v.ray_idx.clone())


@pytest.mark.parametrize("config,B,hw,grid,n_rays,dens", [
    ("bat_blender_VM", 1, (48, 48), 20 ** 3, 300, 22.0),
    ("bat_blender_VM", 2, (40, 56), 24 ** 3, 400, 40.0),
    ("bat_llff_VM_MLP", 1, (36, 48), 9000, 300, 1.0),
    ("bat_llff_VM_MLP", 2, (30, 40), 14000, 500, 3.0),
])
def test_fused_equals_staged_path(config, B, hw, grid, n_rays, dens):
    opt, model, var = _scene(config, B, hw, grid, n_rays, dens)
    se3 = 0.02 * torch.randn(B if False else 1, 6, device=DEV)  # one refinement for the batch, as model/bat.py:268 builds it
    a = _iteration(opt, model, var, se3, fused=False, seed=11)
    b = _iteration(opt, model, var, se3, fused=True, seed=11)
    assert torch.equal(a["ray_idx"], b["ray_idx"])
    # relative to the gradient's max, with a floor: a near-opaque scene leaves a pose gradient of 1e-5, where the staged
    # path's float atomics alone are worth 5e-9
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-4))  # noqa: E731
    e = dict(render=abs(a["render"] - b["render"]) / abs(a["render"]), rgb=float((a["rgb"] - b["rgb"]).abs().max()),
             opacity=float((a["opacity"] - b["opacity"]).abs().max()), depth=float((a["depth"] - b["depth"]).abs().max()),
             g=rel(b["g"], a["g"]))
    print("%s B=%d: fused vs staged: render %.2e rel, rgb %.2e, opacity %.2e, depth %.2e, d/d se3 %.2e of max (|g| max %.3e)"
          % (config, B, e["render"], e["rgb"], e["opacity"], e["depth"], e["g"], float(a["g"].abs().max())))
    assert float(a["opacity"].max()) > 0.3, "the scene must have content"
    assert e["render"] <= 2e-6 and e["rgb"] <= 2e-6 and e["opacity"] <= 2e-6 and e["depth"] <= 2e-5
    assert e["g"] <= 2e-4
    # no atomics on the way to the ray gradients: a second launch gives the same bits
    c = _iteration(opt, model, var, se3, fused=True, seed=11)
    assert torch.equal(b["g"], c["g"]) and torch.equal(b["rgb"], c["rgb"])


def test_fused_equals_oracle_blender():
    """the same iteration in the oracle's stock torch ops (float64 autograd of the restated algorithm would be the
    reference's own arithmetic: this is the fp32 restatement the golden vectors pin)"""
    config, B, hw, grid, n_rays = "bat_blender_VM", 1, (48, 48), 20 ** 3, 300
    opt, model, var = _scene(config, B, hw, grid, n_rays, 22.0)
    g, tf = model.graph, model.graph.nerf.tensorf
    se3 = 0.02 * torch.randn(1, 6, device=DEV)
    b = _iteration(opt, model, var, se3, fused=True, seed=3)
    sd = {k: v.detach().cpu().clone().contiguous() for k, v in tf.state_dict().items()}
    params = O.params_from_state_dict(sd, prefix="")
    cfg = O.SceneCfg(opt.data.scene_bbox, tf.gridSize.tolist(), list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio)
    s = se3.detach().cpu().clone().requires_grad_(True)
    pose = O.compose_pair(O.se3_to_SE3(s), var.pose.cpu())     # eval pose with the identity alignment of _scene
    ray_idx = b["ray_idx"].cpu()
    center, ray = O.rays_for_pixels(pose, var.intr_inv.cpu(), ray_idx, opt.W)
    rgb, depth, acc = O.render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), g.nerf.n_samples, white_bg=True)
    image = var.image.cpu().view(B, 3, -1).permute(0, 2, 1)
    loss = O.render_loss(rgb.view(B, -1, 3), image[:, ray_idx])
    loss.backward()
    e_rgb = float((b["rgb"].cpu().view(-1, 3) - rgb.detach()).abs().max())
    e_g = float((b["g"].cpu() - s.grad).abs().max() / s.grad.abs().max())
    print("fused vs oracle: render %.3e vs %.3e, rgb %.2e, d/d se3 %.2e of max" % (b["render"], float(loss), e_rgb, e_g))
    assert abs(b["render"] - float(loss)) <= 2e-6 * abs(float(loss)) + 1e-9
    assert e_rgb <= 2e-6 and e_g <= 5e-4
