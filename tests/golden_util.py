"""Helpers shared by the parity tests: load a golden fixture and replay it through a renderer."""
import json
import os

import numpy as np
import torch

from oracle import tensorf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = [
    "blender_train_blur", "blender_train_sharp", "blender_vis_blur", "blender_train_mid",
    "blender_train_mid_blur", "blender_train_dense", "blender_train_dense_blur",
    "blender_train_randrays", "llff_train_sharp", "llff_train_blur", "llff_train_thin_whitebg",
    "blender_train_alphamask", "blender_train_shrunk",
    # states taken INSIDE the reference's own training loop (tools/make_engine_trace.py --snapshot, round 6)
    "llff_loop_it21", "llff_loop_it40", "blender_loop_it33",
]


class Fixture:
    def __init__(self, name):
        d = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(d["meta"]).decode())
        self.arrays = {k: d[k] for k in d.files if k != "meta"}

    def t(self, key, device="cpu"):
        return torch.from_numpy(np.array(self.arrays[key])).to(device)

    def has(self, key):
        return key in self.arrays

    def cfg(self, device="cpu"):
        m = self.meta
        return O.SceneCfg(m["aabb"], m["gridSize"], m["near_far"], step_ratio=m["step_ratio"],
                          density_shift=m["density_shift"], distance_scale=m["distance_scale"],
                          fea2denseAct=m["fea2denseAct"], rayMarch_weight_thres=m["rayMarch_weight_thres"],
                          shadingMode=m["shadingMode"], view_pe=m["view_pe"], fea_pe=m["fea_pe"],
                          ndc_near_plane=m["ndc_near_plane"]).to(device)

    def params(self, device="cpu", requires_grad=True):
        sd = {k[len("param."):]: self.t(k, device) for k in self.arrays if k.startswith("param.")}
        p = O.params_from_state_dict(sd)
        if requires_grad:
            for _, v in O.flat_params(p):
                v.requires_grad_(True)
        return p

    def alpha_mask(self, device="cpu"):
        """(volume [D,H,W], aabb [2,3]) of the fixtures rendered with an alpha mask, else None."""
        if not self.has("mask.alpha_volume"):
            return None
        return self.t("mask.alpha_volume", device), self.t("mask.aabb", device)

    def white_bg(self):
        m = self.meta
        coin = m["coin"][0] if m["coin"] else 1.0
        return bool(m["white_bg"] or (m["is_train"] and coin < 0.5))

    def grad_key(self, oracle_name):
        """oracle flat name -> fixture grad key."""
        if oracle_name.startswith("mlp."):
            k = oracle_name[4:]
            idx = {"1": 0, "2": 1, "3": 2}[k[1]]
            kind = "weight" if k[0] == "w" else "bias"
            if "param.nerf.tensorf.renderModule.mlp.0.weight" in self.arrays:
                layer = "renderModule.mlp.%d" % (2 * idx)
            else:
                layer = "renderModule.layer%d" % (idx + 1)
            return "grad.nerf.tensorf.%s.%s" % (layer, kind)
        return "grad.nerf.tensorf." + oracle_name


def replay_oracle(fx, use_taps=False, device="cpu", pin_rays=True):
    """Run the whole path of one fixture through the oracle; returns dict of outputs and grads.

    pin_rays: substitute the VALUES of the reference's (center, ray_dir) for the recomputed ones while
    keeping the autograd graph.  Without jitter the first sample of every ray sits exactly on the AABB
    face (z = t_min), so the in-box test is decided by the last bit of the ray; pinning removes that
    coin toss from the comparison (a + (b - a) == b exactly for nearby floats)."""
    m = fx.meta
    cfg = fx.cfg(device)
    params = fx.params(device)
    se3 = fx.t("param.se3_refine.weight", device).clone().requires_grad_(True)
    idx = fx.t("in.idx", device).long()
    pose_gt = fx.t("in.pose_gt", device)
    if m["llff"]:
        pose = O.train_pose(se3[idx], None, torch.eye(3, 4, device=device))
    else:
        pose = O.train_pose(se3[idx], fx.t("param.pose_noise", device)[idx], pose_gt)
    ray_idx = fx.t("in.ray_idx", device).long()
    center, ray = O.rays_for_pixels(pose, fx.t("in.intr_inv", device), ray_idx, m["W"])
    if m["ndc_ray"]:
        center, ray = O.convert_ndc(center, ray, fx.t("in.intr", device), near=m["ndc_near_plane"])
    B, r = center.shape[:2]
    center_own, ray_own = center, ray
    if pin_rays:
        center = center + (fx.t("mid.center", device).view(B, r, 3) - center).detach()
        ray = ray + (fx.t("mid.ray_dir", device).view(B, r, 3) - ray).detach()
    kd = kc = None
    if m["c2f_mode"] is not None:
        kd = O.get_kernel(cfg, m["c2f_parameter_density"], m["c2f_kernel_size"]).to(device)
        kc = O.get_kernel(cfg, m["c2f_parameter_color"], m["c2f_kernel_size"]).to(device)
    jitter = fx.t("in.jitter", device) if (m["is_train"] and fx.has("in.jitter")) else None
    rgb, depth, acc, aux = O.render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), m["N_samples"],
                                    white_bg=fx.white_bg(), jitter=jitter, ndc_ray=m["ndc_ray"],
                                    kernel_density=kd, kernel_color=kc,
                                    view_pe_progress=m["view_pe_progress"], fea_pe_progress=m["fea_pe_progress"],
                                    use_taps=use_taps, return_aux=True, alpha_mask=fx.alpha_mask(device))
    rgb = rgb.view(B, r, 3)
    out = dict(pose=pose, center=center_own, ray=ray_own, rgb=rgb, depth=depth.view(B, r, 1), opacity=acc.view(B, r, 1),
               kd=kd, kc=kc, aux=aux)
    # losses
    losses = {}
    if m["mode"] == "vis":
        losses["render"] = ((rgb - 0.3) ** 2).mean()
        total = losses["render"]
    else:
        image = fx.t("in.image", device).view(B, 3, -1).permute(0, 2, 1)[:, ray_idx]
        e = m["edge_loss"]
        edge_on = e["on"] and ((m["it"] % 2 == 0) if e["alternate"] else True)
        if edge_on and m["mode"] == "train" and m["it"] < e["before_iter"]:
            mask = fx.t("in.train_edge_masks", device)[:, ray_idx]
            losses["render"] = O.render_loss(rgb, image, mask, e["edge_factor"], e["non_edge_factor"])
        else:
            losses["render"] = O.render_loss(rgb, image)
        losses["L1"] = O.density_L1(params)
        losses["TV_density"] = O.tv_planes(params["density_plane"])
        losses["TV_color"] = O.tv_planes(params["app_plane"])
        total = losses["render"] + m["L1_weight"] * losses["L1"] + m["TV_density_weight"] * losses["TV_density"] \
            + m["TV_color_weight"] * losses["TV_color"]
        if "loss.TV_depth" in fx.arrays:
            losses["TV_depth"] = O.tv_depth(depth, B, m["grid_H"], m["grid_W"])
    total.backward()
    out["losses"] = losses
    out["total"] = total
    out["grads"] = {n: v.grad for n, v in O.flat_params(params)}
    out["grad_se3"] = se3.grad
    out["params"] = params
    return out
