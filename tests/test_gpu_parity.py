"""GPU parity: the HIP path (through the drop-in scene class and the C ABI) against
  (a) the golden vectors captured from the reference, and
  (b) the CPU oracle on the same inputs.
Tolerances = ~4x the errors measured on MI355X (printed per case; round 2: rgb / opacity <= 4.8e-7, gradients
<= 8.7e-5 of the tensor's max-abs on the Blender fixtures, <= 2.1e-4 on the LLFF ones, 2.1e-3 on the saturated
`dense` fixtures whose render gradient is a cancellation residue): values 2e-6 abs, gradients 5e-4 / 1e-3 (LLFF) / 8e-3.
The full-size configurations are in tests/test_gpu_fullsize.py."""
import numpy as np
import pytest
import torch

from tests.golden_util import CASES, Fixture, replay_oracle

pytestmark = pytest.mark.gpu

TOL_VAL = 2e-6
TOL_GRAD = 5e-4


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build_scene(fx, device, shade_impl="mfma"):
    import joint_tensorf_amd as jt
    m = fx.meta
    torch.manual_seed(0)
    tf = jt.BAT_VMSplit(m["aabb"], m["gridSize"], device, density_n_comp=m["density_n_comp"],
                        appearance_n_comp=m["app_n_comp"], app_dim=m["app_dim"], near_far=list(m["near_far"]),
                        shadingMode=m["shadingMode"], density_shift=m["density_shift"],
                        distance_scale=m["distance_scale"], view_pe=m["view_pe"], fea_pe=m["fea_pe"],
                        featureC=m["featureC"], step_ratio=m["step_ratio"], fea2denseAct=m["fea2denseAct"],
                        rayMarch_weight_thres=m["rayMarch_weight_thres"])
    sd = {k[len("param.nerf.tensorf."):]: fx.t(k) for k in fx.arrays if k.startswith("param.nerf.tensorf.")}
    missing, unexpected = tf.load_state_dict(sd, strict=True)
    tf = tf.to(device)
    if fx.has("mask.alpha_volume"):  # fixtures rendered with an alpha mask
        from joint_tensorf_amd.tensorf_repr import AlphaGridMask
        tf.alphaMask = AlphaGridMask(device, fx.t("mask.aabb", device), fx.t("mask.alpha_volume", device))
    return tf


def replay_hip(fx, shade_impl, pin_rays=True, device="cuda"):
    from joint_tensorf_amd import ops
    m = fx.meta
    tf = build_scene(fx, device, shade_impl)
    se3 = fx.t("param.se3_refine.weight", device).clone().requires_grad_(True)
    idx = fx.t("in.idx", device).long()
    if m["llff"]:
        pose = ops.train_pose(se3[idx], None, torch.eye(3, 4, device=device))
    else:
        pose = ops.train_pose(se3[idx], fx.t("param.pose_noise", device)[idx], fx.t("in.pose_gt", device))
    ray_idx = fx.t("in.ray_idx", device).long()
    center, ray = ops.ray_gen(pose, fx.t("in.intr_inv", device), fx.t("in.intr", device), ray_idx, m["W"],
                              ndc=m["ndc_ray"], ndc_near=m["ndc_near_plane"])
    B, r = center.shape[:2]
    center_own, ray_own = center, ray
    if pin_rays:
        center = center + (fx.t("mid.center", device).view(B, r, 3) - center).detach()
        ray = ray + (fx.t("mid.ray_dir", device).view(B, r, 3) - ray).detach()
    if m["is_train"] and fx.has("in.jitter"):
        tf.jitter_override = fx.t("in.jitter", device)
    tf.coin_override = m["coin"][0] if m["coin"] else None
    import contextlib
    from tests.staged_path import use_staged_path
    # "torch": the staged cross-check path (tests/staged_path.py) instead of the fused MFMA kernels
    with (use_staged_path() if shade_impl == "torch" else contextlib.nullcontext()):
        rgb, depth, acc = tf(None, center.reshape(-1, 3), ray.reshape(-1, 3), white_bg=m["white_bg"],
                             is_train=m["is_train"], ndc_ray=m["ndc_ray"], N_samples=m["N_samples"],
                             c2f_parameter_density=m["c2f_parameter_density"], c2f_parameter_color=m["c2f_parameter_color"],
                             c2f_mode=m["c2f_mode"], c2f_kernel_size=m["c2f_kernel_size"],
                             view_pe_progress=m["view_pe_progress"], fea_pe_progress=m["fea_pe_progress"])
    rgb = rgb.view(B, r, 3)
    from oracle import tensorf_oracle as O  # checker only
    if m["mode"] == "vis":
        total = ((rgb - 0.3) ** 2).mean()
    else:
        image = fx.t("in.image", device).view(B, 3, -1).permute(0, 2, 1)[:, ray_idx]
        e = m["edge_loss"]
        edge_on = e["on"] and ((m["it"] % 2 == 0) if e["alternate"] else True)
        if edge_on and m["mode"] == "train" and m["it"] < e["before_iter"]:
            mask = fx.t("in.train_edge_masks", device)[:, ray_idx]
            render = O.render_loss(rgb, image, mask, e["edge_factor"], e["non_edge_factor"])
        else:
            render = O.render_loss(rgb, image)
        import joint_tensorf_amd as jt
        tv = jt.TVLoss()
        total = render + m["L1_weight"] * tf.density_L1() + m["TV_density_weight"] * tf.TV_loss_density(tv) \
            + m["TV_color_weight"] * tf.TV_loss_app(tv)
    total.backward()
    grads = {}
    for grp in ("density_plane", "density_line", "app_plane", "app_line"):
        for i in range(3):
            grads["%s.%d" % (grp, i)] = getattr(tf, grp)[i].grad
    grads["basis_mat.weight"] = tf.basis_mat.weight.grad
    w = tf.renderModule.weights()
    for k, t in zip(("w1", "b1", "w2", "b2", "w3", "b3"), w):
        grads["mlp." + k] = t.grad
    return dict(pose=pose, center=center_own, ray=ray_own, rgb=rgb, depth=depth.view(B, r, 1),
                opacity=acc.view(B, r, 1), total=total, grads=grads, grad_se3=se3.grad)


# kernel variants of the appearance path every fixture is held to: the default (bf16x3 forward / weight gradients, fused
# backward), the staged cross-check path, the fp32-matrix-core kernels (jt_shade_set_matrix_mode(0): VERDICT r3 "thin spot")
# and the split backward (chain kernel + scatter kernel with runs of 16 / 8 samples, jt_shade_set_bwd_split)
# (matrix mode, backward split; -1 = the library's per-scene default: fused for VM-48, split 16 for the 20-channel scene)
# (matrix-mode bit 2, round 5: the chain of a SPLIT backward on the bf16 matrix cores -- 7 = the library's default)
# (third entry, round 6: the lean tape -- no product records, dBasis out of the scatter kernel; it only takes effect where the
#  backward is the split form with the walker scatter.  "mfma-fulltape" is round 5's default: product records + dBasis GEMM)
VARIANTS = {"mfma": (7, -1, 1), "torch": (7, -1, 1), "mfma-fp32": (0, 0, 1), "mfma-split16": (7, 16, 1),
            "mfma-split8-fp32": (0, 8, 1), "mfma-tile": (7, 1, 1), "mfma-split16-fp32chain": (3, 16, 1),
            "mfma-fulltape": (7, -1, 0)}


class kernel_variant:
    """with kernel_variant(name): the library's matrix mode / backward split for the launches inside, restored on exit"""

    def __init__(self, name):
        self.mode, self.split, self.lean = VARIANTS[name]

    def __enter__(self):
        from joint_tensorf_amd._lib import lib
        self.prev = (lib.jt_shade_set_matrix_mode(self.mode), lib.jt_shade_set_bwd_split(self.split),
                     lib.jt_shade_set_lean_tape(self.lean))
        assert lib.jt_shade_matrix_mode() == self.mode and lib.jt_shade_bwd_split() == self.split
        assert lib.jt_shade_lean_tape() == self.lean
        return self

    def __exit__(self, *exc):
        from joint_tensorf_amd._lib import lib
        lib.jt_shade_set_matrix_mode(self.prev[0])
        lib.jt_shade_set_bwd_split(self.prev[1])
        lib.jt_shade_set_lean_tape(self.prev[2])
        return False


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("shade_impl", list(VARIANTS))
def test_hip_vs_golden_and_oracle(name, shade_impl):
    fx = Fixture(name)
    with kernel_variant(shade_impl):
        out = replay_hip(fx, "torch" if shade_impl == "torch" else "mfma")
    ref = replay_oracle(fx)
    tol_g = 8e-3 if "dense" in name else (1e-3 if name.startswith("llff") else TOL_GRAD)  # ~4x the measured worst
    np.testing.assert_allclose(out["pose"].detach().cpu().numpy(), fx.arrays["mid.current_pose"], atol=2e-6)
    np.testing.assert_allclose(out["center"].detach().cpu().reshape(-1, 3).numpy(), fx.arrays["mid.center"], atol=5e-6)
    np.testing.assert_allclose(out["ray"].detach().cpu().reshape(-1, 3).numpy(), fx.arrays["mid.ray_dir"], atol=5e-6)
    for key, gold in (("rgb", "out.rgb"), ("opacity", "out.opacity")):
        np.testing.assert_allclose(out[key].detach().cpu().numpy(), fx.arrays[gold], atol=TOL_VAL, err_msg=key)
        np.testing.assert_allclose(out[key].detach().cpu().numpy(), ref[key].detach().numpy(), atol=TOL_VAL, err_msg=key)
    np.testing.assert_allclose(out["depth"].detach().cpu().numpy(), fx.arrays["out.depth"], atol=1e-4)
    if fx.meta["mode"] != "vis":
        np.testing.assert_allclose(float(out["total"].detach()), float(fx.arrays["loss.all"]), rtol=1e-4)
    bad, worst = [], ("", 0.0)
    for n, g in out["grads"].items():
        key = fx.grad_key(n)
        assert g is not None, n
        e1 = _rel(g.detach().cpu().numpy(), fx.arrays[key])
        e2 = _rel(g.detach().cpu().numpy(), ref["grads"][n].numpy())
        if max(e1, e2) > tol_g:
            bad.append((n, e1, e2))
        if max(e1, e2) > worst[1]:
            worst = (n, max(e1, e2))
    e = _rel(out["grad_se3"].cpu().numpy(), fx.arrays["grad.se3_refine.weight"])
    if e > tol_g:
        bad.append(("se3", e, _rel(out["grad_se3"].cpu().numpy(), ref["grad_se3"].numpy())))
    ev = max(float(np.abs(out[k].detach().cpu().numpy() - fx.arrays[g_]).max()) for k, g_ in (("rgb", "out.rgb"), ("opacity", "out.opacity")))
    print("\n[parity] %-28s %-5s rgb/opacity %.1e  worst gradient %s %.1e  se3 %.1e" % (name, shade_impl, ev, worst[0], worst[1], e))
    assert not bad, bad
