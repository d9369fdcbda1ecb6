"""VMAdam: torch.optim.Adam's arithmetic, state layout and param-group interface (model/tensorf.py:463-478 builds
`torch.optim.Adam(grad_vars, betas=(0.9, 0.99))`) with the step of ALL parameter tensors in one HIP launch
(jt_adam_step).  State keys are Adam's (`step`, `exp_avg`, `exp_avg_sq`), so optimizer
checkpoints interchange with torch.optim.Adam."""
import math

import torch

from ._lib import JtAdamItem, check, lib, ptr
from .ops import _stream


class VMAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # one launch per (betas, eps) combination; the reference uses a single one
        batches = {}
        keep = []
        for group in self.param_groups:
            key = (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("VMAdam steps parameters on the GPU only (no CPU fallback)")
                if g.is_sparse or p.dtype != torch.float32:
                    raise RuntimeError("VMAdam: dense float32 parameters only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                if g.stride() != p.stride() or g.dtype != torch.float32:
                    # the kernel walks all four tensors in the parameter's memory order
                    g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                    g2.copy_(g)
                    g = g2
                    keep.append(g)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                assert m.stride() == p.stride() and v.stride() == p.stride()
                batches.setdefault(key, []).append(
                    (p, g, m, v, float(group["lr"]), 1.0 - key[0] ** t, 1.0 - key[1] ** t))
        stream = _stream()
        for (b1, b2, eps), items in batches.items():
            arr = (JtAdamItem * len(items))()
            for k, (p, g, m, v, lr, bc1, bc2) in enumerate(items):
                arr[k].p, arr[k].g, arr[k].m, arr[k].v = ptr(p), ptr(g), ptr(m), ptr(v)
                arr[k].n, arr[k].lr, arr[k].bias_correction1, arr[k].bias_correction2 = p.numel(), lr, bc1, bc2
            check(lib.jt_adam_step(arr, len(items), b1, b2, eps, stream), "jt_adam_step")
        return loss
