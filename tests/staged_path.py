"""Cross-check aid (test infrastructure, not product): the SAME march / composite kernels around a STAGED appearance
path -- jt_app_gather_forward materialises the [n, 3*Ca] plane x line products, the basis / positional encoding / MLP
run in stock torch ops (torch autograd for their backward), jt_app_gather_backward scatters the product gradients.
tests/test_gpu_parity.py runs every fixture through it next to the fused MFMA path (ops.RenderRays): two independent
implementations of the appearance chain against the same golden vectors.  One host sync per forward (the shaded count)."""
import ctypes
import os

import torch

from joint_tensorf_amd import _lib, ops
from joint_tensorf_amd._lib import FP, I, P, SP, check, lib, ptr

# the staged path's own two kernels live in a TEST-ONLY library (tests/csrc/jt_app.hip, built next to the product library
# by joint_tensorf_amd/build.py); everything else -- march, composite, density backward -- is the product library's
_TEST_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libjt_test_staged.so")
if not os.path.exists(_TEST_LIB):
    raise ImportError("%s is missing: run `python joint_tensorf_amd/build.py`" % _TEST_LIB)
tlib = ctypes.CDLL(_TEST_LIB)
tlib.jt_app_gather_forward.restype = I
tlib.jt_app_gather_forward.argtypes = [SP, FP, P, P, P, P, P, P, I, P, P, P, I, P]
tlib.jt_app_gather_backward.restype = I
tlib.jt_app_gather_backward.argtypes = [SP, FP, P, P, P, P, P, P, I, P, P, P, FP, P, I, P]
from joint_tensorf_amd.ops import _factors_struct, _stream, factor_logical, factor_storage


def _pe(x, freqs, progress):
    levels = torch.arange(freqs, device=x.device)
    bands = (2 ** levels).to(x.dtype)
    mask = (progress * freqs - levels).clamp(0.0, 1.0).to(x.dtype)
    pts = x[..., None] * bands
    pts = torch.cat([torch.sin(pts) * mask, torch.cos(pts) * mask], -1)
    return pts.reshape(x.shape[:-1] + (freqs * 2 * x.shape[-1],))


def _torch_shade(cfg, prod, vdir, basis, w1, b1, w2, b2, w3, b3):
    feat = prod @ basis.t()
    F = torch.nn.functional
    if cfg.mlp_kind == _lib.JT_MLP_WEAKVIEW:
        x = torch.cat([feat, _pe(feat, cfg.fea_pe, cfg.fea_pe_progress)], -1) if cfg.fea_pe > 0 else feat
        h = F.relu(F.linear(x, w1, b1))
        h = F.relu(F.linear(h, w2, b2))
        mid = torch.cat([_pe(vdir, cfg.view_pe, cfg.view_pe_progress), h], -1) if cfg.view_pe > 0 else h
        return torch.sigmoid(F.linear(mid, w3, b3))
    x = [feat, vdir]
    if cfg.fea_pe > 0:
        x.append(_pe(feat, cfg.fea_pe, cfg.fea_pe_progress))
    if cfg.view_pe > 0:
        x.append(_pe(vdir, cfg.view_pe, cfg.view_pe_progress))
    h = F.relu(F.linear(torch.cat(x, -1), w1, b1))
    h = F.relu(F.linear(h, w2, b2))
    return torch.sigmoid(F.linear(h, w3, b3))


class StagedRenderRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, rays_o, rays_d, jitter, zvals, *params):
        dp, dl, ap, al = params[0:3], params[3:6], params[6:9], params[9:12]
        dev = rays_o.device
        R, S = rays_o.shape[0], cfg.n_samples
        rays_o = rays_o.detach().contiguous().float()
        rays_d = rays_d.detach().contiguous().float()
        jitter = None if jitter is None else jitter.detach().contiguous().float().view(-1)
        zvals = None if zvals is None else zvals.detach().contiguous().float().view(-1)
        sdp, sdl, sap, sal = ([factor_storage(p) for p in lst] for lst in (dp, dl, ap, al))
        mlp_t = [t.detach().contiguous() for t in params[12:19]]
        scene = cfg.scene()
        fac = _factors_struct(sdp, sdl, sap, sal, cfg.alpha_mask[0] if cfg.alpha_mask is not None else None)
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        sigma_feat, weight, tmin = torch.empty(R, S, **f32), torch.empty(R, S, **f32), torch.empty(R, **f32)
        count = torch.empty(R, device=dev, dtype=torch.int32)
        offset = torch.empty(R + 1, device=dev, dtype=torch.int32)
        sidx = torch.empty(R, S, device=dev, dtype=torch.int16)
        opacity, depth = torch.empty(R, **f32), torch.empty(R, **f32)
        check(lib.jt_march_forward(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), R, ptr(sigma_feat),
                                   ptr(weight), ptr(tmin), ptr(count), ptr(offset), ptr(sidx), ptr(opacity), ptr(depth),
                                   st), "jt_march_forward")
        n = cap = int(offset[R].item())
        cap_alloc = max(cap, 1)
        eray = torch.empty(cap_alloc, device=dev, dtype=torch.int32)
        esmp = torch.empty(cap_alloc, device=dev, dtype=torch.int32)
        vdir = torch.empty(cap_alloc, 3, **f32)
        check(lib.jt_shade_list(scene, ptr(rays_d), R, ptr(offset), ptr(sidx), ptr(eray), ptr(esmp), ptr(vdir), cap, st),
              "jt_shade_list")
        rgb_s = torch.empty(cap_alloc, 3, **f32)
        prod = torch.empty(cap_alloc, 3 * cfg.n_comp_app, **f32)
        check(tlib.jt_app_gather_forward(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), ptr(tmin),
                                        ptr(offset), R, ptr(eray), ptr(esmp), ptr(prod), cap, st), "jt_app_gather_forward")
        if n > 0:
            with torch.no_grad():
                rgb_s[:n] = _torch_shade(cfg, prod[:n], vdir[:n], *mlp_t)
        rgb = torch.empty(R, 3, **f32)
        cmask = torch.empty(R, device=dev, dtype=torch.int32)
        check(lib.jt_composite_forward(scene, R, ptr(offset), ptr(sidx), ptr(weight), ptr(rgb_s), ptr(opacity), ptr(rgb),
                                       ptr(cmask), st), "jt_composite_forward")
        ctx.cfg, ctx.n, ctx.cap, ctx.prod = cfg, n, cap, prod
        ctx.saved = (rays_o, rays_d, jitter, zvals, sdp, sdl, sap, sal, mlp_t, sigma_feat, weight, tmin, offset, sidx,
                     eray, esmp, vdir, rgb_s, cmask)
        ctx.mark_non_differentiable(depth)
        ctx.set_materialize_grads(False)
        cfg.shade_lists = (offset, sidx)
        return rgb, depth, opacity

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opacity):
        prev = lib.jt_set_deterministic(0)  # float gradient buffers: the fixed-point mode of the product path is off here
        try:
            return StagedRenderRays._backward(ctx, g_rgb, g_depth, g_opacity)
        finally:
            lib.jt_set_deterministic(prev)

    @staticmethod
    def _backward(ctx, g_rgb, g_depth, g_opacity):
        cfg = ctx.cfg
        (rays_o, rays_d, jitter, zvals, sdp, sdl, sap, sal, mlp_t, sigma_feat, weight, tmin, offset, sidx, eray, esmp,
         vdir, rgb_s, cmask) = ctx.saved
        dev, R = rays_o.device, rays_o.shape[0]
        scene = cfg.scene()
        fac = _factors_struct(sdp, sdl, sap, sal, cfg.alpha_mask[0] if cfg.alpha_mask is not None else None)
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        g_rgb = torch.zeros(R, 3, **f32) if g_rgb is None else g_rgb.contiguous().float()
        g_op = None if g_opacity is None else g_opacity.contiguous().float()
        cap, n = ctx.cap, ctx.n
        cap_alloc = max(cap, 1)
        g_rgb_s = torch.empty(cap_alloc, 3, **f32)
        check(lib.jt_composite_backward(scene, R, ptr(offset), ptr(eray), ptr(esmp), ptr(weight), ptr(cmask), ptr(g_rgb),
                                        ptr(g_rgb_s), cap, st), "jt_composite_backward")
        gdp, gdl, gap, gal = ([torch.zeros_like(t) for t in lst] for lst in (sdp, sdl, sap, sal))
        gfac = _factors_struct(gdp, gdl, gap, gal)
        g_xyz = torch.empty(cap_alloc, 3, **f32)
        g_mlp = [torch.zeros_like(t) for t in mlp_t]
        if n > 0:
            prod = ctx.prod[:n].detach().requires_grad_(True)
            leaves = [t.detach().requires_grad_(True) for t in mlp_t]
            with torch.enable_grad():
                out = _torch_shade(cfg, prod, vdir[:n], *leaves)
            grads = torch.autograd.grad(out, [prod] + leaves, g_rgb_s[:n])
            g_prod, g_mlp = grads[0].contiguous(), list(grads[1:])
        else:
            g_prod = torch.zeros(1, 3 * cfg.n_comp_app, **f32)
        check(tlib.jt_app_gather_backward(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), ptr(tmin),
                                         ptr(offset), R, ptr(eray), ptr(esmp), ptr(g_prod), gfac, ptr(g_xyz), cap, st),
              "jt_app_gather_backward")
        g_o, g_d = torch.empty(R, 3, **f32), torch.empty(R, 3, **f32)
        mws_bytes = lib.jt_march_backward_workspace_bytes(scene, R)
        mws = torch.empty(max(int(mws_bytes), 16), device=dev, dtype=torch.uint8)
        check(lib.jt_march_backward(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), R, ptr(sigma_feat),
                                    ptr(weight), ptr(tmin), ptr(offset), ptr(sidx), ptr(rgb_s), ptr(cmask), ptr(g_rgb),
                                    ptr(g_op), ptr(g_xyz), gfac, ptr(g_o), ptr(g_d), ptr(mws), mws_bytes, st),
              "jt_march_backward")
        g_factors = [factor_logical(t) for t in gdp + gdl + gap + gal]
        return tuple([None, g_o, g_d, None, None] + g_factors + list(g_mlp))


def staged_render_rays(cfg, rays_o, rays_d, jitter, zvals, density_plane, density_line, app_plane, app_line, basis,
                       mlp_params):
    cfg.reg3 = None
    cfg.reg_flags = None
    return StagedRenderRays.apply(cfg, rays_o, rays_d, jitter, zvals, *density_plane, *density_line, *app_plane, *app_line,
                                  basis, *mlp_params)


class use_staged_path:
    """`with use_staged_path():` -- BAT_VMSplit.forward renders through the staged appearance path"""

    def __enter__(self):
        self.orig = ops.render_rays
        ops.render_rays = staged_render_rays
        return self

    def __exit__(self, *exc):
        ops.render_rays = self.orig
        return False
