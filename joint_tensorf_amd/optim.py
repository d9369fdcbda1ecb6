"""VMAdam: torch.optim.Adam's arithmetic, state layout and param-group interface (model/tensorf.py:463-478 builds
`torch.optim.Adam(grad_vars, betas=(0.9, 0.99))`) with the step of ALL parameter tensors in one HIP launch
(jt_adam_step).  State keys are Adam's (`step`, `exp_avg`, `exp_avg_sq`), so optimizer
checkpoints interchange with torch.optim.Adam."""
import math
import os

import torch

from ._lib import JtAdamItem, check, lib, ptr
from .ops import _stream

# JT_ADAM_PLAN=0: every eager step through the general path (the planned step is the same launch with the same arguments)
PLAN_STEPS = os.environ.get("JT_ADAM_PLAN", "1") != "0"


def _same_layout(a, b):
    """same memory order: equal strides on every dimension that has more than one element (the stride of a size-1
    dimension is arbitrary -- the sum autograd forms of two channel-last gradients carries a different one there)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n != 1)


class VMAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict keeps the saved tensors' strides.  A checkpoint written by the reference
        (torch.optim.Adam over contiguous NCHW factors, util.py:160-184) -- or by this build's save_checkpoint, which
        stores plain contiguous tensors -- is then laid out differently from the channel-last parameters the step kernel
        walks: every moment tensor is re-laid into its parameter's memory order by value."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st and torch.is_tensor(st[k]) and not _same_layout(st[k], p):
                        st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[k])
        self.__dict__.pop("_layout_memo", None)
        self.__dict__.pop("_plan", None)

    # The step is split in two so that a hipGraph can hold the launch (graphed.GraphedTrainStep): `prepare_step`
    # is the host half -- Adam's step counters, lr and bias corrections in Python doubles exactly as
    # torch.optim.Adam computes them -- and ends with ONE jt_poke that puts the coefficients of all tensors into
    # device memory; `launch_step` is the device half and reads them from there.

    def _items(self):
        """[(p, group)] of every parameter with a gradient, grouped by (betas, eps), in param-group order."""
        batches = {}
        for group in self.param_groups:
            key = (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
            for p in group["params"]:
                if p.grad is None:
                    continue
                batches.setdefault(key, []).append((p, group))
        return batches

    def _layout_ok(self, a, p):
        """_same_layout(a, p), remembered per (parameter, shape and strides of the other tensor)."""
        memo = self.__dict__.setdefault("_layout_memo", {})
        key = (id(p), a.shape, a.stride(), p.stride())
        r = memo.get(key)
        if r is None:
            if len(memo) > 4096:
                memo.clear()
            r = memo[key] = _same_layout(a, p)
        return r

    def _dyn_buffer(self, dev, n_items):
        d = self.__dict__.setdefault("_dyn", {})
        from .ops import device_key
        key = device_key(dev)   # ("cuda" before a capture, "cuda:0" inside it: one buffer)
        if key not in d or d[key].numel() < 2 * n_items:
            d[key] = torch.zeros(2 * max(n_items, 32), device=torch.device("cuda", key), dtype=torch.float32)
            self._dyn_gen = getattr(self, "_dyn_gen", 0) + 1  # a captured graph holding the old buffer is stale
        return d[key]

    @torch.no_grad()
    def prepare_step(self, params_with_grad=None, poke=True):
        """Advance the step counters and write this iteration's (lr / bias_correction1, 1 / sqrt(bias_correction2))
        of every tensor to the device (poke=True: one jt_poke, for a launch that reads them from device memory -- a captured
        hipGraph) or just return them (poke=False: the eager step hands them to the launch as arguments).
        `params_with_grad`: the ids of the parameters the following launch will step (default: those that have a .grad now);
        the order must be the one `launch_step` sees."""
        coefs = []
        dev = None
        for group in self.param_groups:
            b1, b2 = float(group["betas"][0]), float(group["betas"][1])
            for p in group["params"]:
                if (p.grad is None) if params_with_grad is None else (id(p) not in params_with_grad):
                    continue
                if not p.is_cuda:
                    raise RuntimeError("VMAdam steps parameters on the GPU only (no CPU fallback)")
                dev = p.device
                st = self.state[p]
                if len(st) == 0:
                    self._state_created = True   # (zero fills on THIS stream just now: see step)
                    st["step"] = 0.0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # `step` is kept as a Python float (a 0-dim tensor loaded from a torch.optim.Adam checkpoint is
                # converted on first use; torch's Adam converts the other way in __setstate__): 19 tensor updates and
                # read-backs per iteration are 40 us of host time
                t = float(st["step"]) + 1.0
                st["step"] = t
                coefs.append(float(group["lr"]) / (1.0 - b1 ** t))
                coefs.append(1.0 / math.sqrt(1.0 - b2 ** t))
        if not coefs:
            return coefs
        if len({(float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for g in self.param_groups}) != 1:
            raise RuntimeError("VMAdam: one (betas, eps) combination per optimizer (the reference uses a single one)")
        # (the buffer exists from the first step on, poked or not: a hipGraph capture that meets launch_step() must find it --
        #  allocated inside the capture, its zero fill would be replayed behind every poke)
        dyn = self._dyn_buffer(dev, len(coefs) // 2)
        if poke:
            from .ops import poke_floats
            for lo in range(0, len(coefs), 256):
                poke_floats(dyn, coefs[lo:lo + 256], offset=lo)
        return coefs

    @torch.no_grad()
    def launch_step(self, coefs=None, only=None, stream=None):
        """One launch over all parameter tensors that have a gradient.  coefs = None: the coefficients `prepare_step` poked
        into device memory; a list: the same values as launch arguments (same floats, same arithmetic in the kernel).
        only = (addresses, keep): just the tensors whose GRADIENT's address is (keep = True) / is not (False) among them, with
        their own coefficients (coefs as a list only); stream: raw handle of the stream to launch on (default: the current)."""
        keep = []
        stream = _stream() if stream is None else stream
        for (b1, b2, eps), plist in self._items().items():
            if only is not None:
                sel = [k for k, (p, _) in enumerate(plist) if (int(p.grad.data_ptr()) in only[0]) == only[1]]
                if not sel:
                    continue
                plist = [plist[k] for k in sel]
                coefs = [c for k in sel for c in coefs[2 * k:2 * k + 2]]
            arr = (JtAdamItem * len(plist))()
            for k, (p, group) in enumerate(plist):
                g = p.grad
                if g.is_sparse or p.dtype != torch.float32:
                    raise RuntimeError("VMAdam: dense float32 parameters only")
                st = self.state[p]
                if g.dtype != torch.float32 or not self._layout_ok(g, p):
                    # the kernel walks all four tensors in the parameter's memory order
                    g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                    g2.copy_(g)
                    g = g2
                    keep.append(g)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                assert self._layout_ok(m, p) and self._layout_ok(v, p)
                arr[k].p, arr[k].g, arr[k].m, arr[k].v, arr[k].n = ptr(p), ptr(g), ptr(m), ptr(v), p.numel()
            if coefs is not None:
                import ctypes
                host = (ctypes.c_float * len(coefs))(*coefs)
                check(lib.jt_adam_step_coefs(arr, len(plist), b1, b2, eps, host, stream), "jt_adam_step_coefs")
            else:
                dyn = self._dyn_buffer(plist[0][0].device, len(plist))
                check(lib.jt_adam_step_dyn(arr, len(plist), b1, b2, eps, ptr(dyn), stream), "jt_adam_step_dyn")

    supports_early_step = True

    @torch.no_grad()
    def step(self, closure=None, early_ok=False):
        """early_ok: only a caller that guarantees NOTHING touches the parameters' .grad between the render
        backward and this call -- Model.end_iteration, the bench step that mirrors it -- may let the appearance factors be stepped
        on the auxiliary stream behind the event the backward recorded in its middle (ops.RenderRays.backward's offer).  Gradient
        clipping or scaling in place, a hook, a later AccumulateGrad add keep data_ptr unchanged: a plain step() therefore never
        takes the offer and runs as one launch on the current stream, after everything the caller enqueued."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if PLAN_STEPS and self._planned_step(early_ok):
            return loss
        coefs = self.prepare_step(poke=False)
        if not coefs:
            return loss
        from . import ops
        early = None
        # (a step that has just created moments filled them on this stream a moment ago -- behind the event the auxiliary
        #  stream would wait for: that step stays on this stream as a whole)
        fresh = self.__dict__.pop("_state_created", False)
        if ops._EARLY_GRADS and not early_ok:
            ops._EARLY_GRADS.clear()   # (an offer nobody may take: dropped, so that a later step cannot pick up a stale one)
        if early_ok and ops._EARLY_GRADS and not fresh and len(self._items()) == 1:
            dev = next(p for g in self.param_groups for p in g["params"] if p.grad is not None).device
            offer = ops._EARLY_GRADS.get(ops.device_key(dev))
            # mine = the offered gradients that ARE the .grad of one of my tensors, in the layout the kernel walks (a gradient
            # that autograd copied, or one that needs a layout copy first, stays on this stream; the pose optimizer has none)
            mine = frozenset(int(p.grad.data_ptr()) for g in self.param_groups for p in g["params"]
                             if p.grad is not None and offer is not None and int(p.grad.data_ptr()) in offer[1]
                             and p.grad.dtype == torch.float32 and self._layout_ok(p.grad, p))
            if mine:
                early = ops.take_early_grads(dev)
        if early is None:
            self.launch_step(coefs)
            self._make_plan(None, None)
            return loss
        # The tensors whose gradients the render backward declared final behind its appearance half (ops.RenderRays.backward)
        # are stepped on the auxiliary stream from that point on, i.e. beside the density backward, which waits on the
        # float-atomic path while this launch streams parameters, moments and gradients; everything else on this stream as
        # before, and this stream goes on only when both are done.
        ev, _, storage, aux = early
        addresses = mine
        aux.wait_event(ev)
        storage.record_stream(aux)
        self.launch_step(coefs, only=(addresses, True), stream=aux.cuda_stream)
        done = self.__dict__.setdefault("_early_done", {}).setdefault(ops.device_key(dev), torch.cuda.Event())
        done.record(aux)
        self.launch_step(coefs, only=(addresses, False))
        torch.cuda.current_stream().wait_event(done)
        self.early_steps = getattr(self, "early_steps", 0) + 1
        self._make_plan(offer[1], addresses)
        return loss

    # ---- the steady state: an eager step whose tensors are where they were one step ago ---------------------------------------
    # Walking param_groups three times, building the launch's item array and checking every tensor's layout is ~0.2 ms of host
    # time per step for the 22 tensors of the scene -- as much as a host-bound iteration spends in all its forward launches.
    # After a step that went through the general path above, `_make_plan` keeps what that step established: the item arrays
    # (one, or the early / late pair), and per tensor what was CHECKED to get there (addresses of parameter and gradient, the
    # gradient's strides and dtype, the identity of the moment tensors).  `_planned_step` re-verifies exactly those facts and, when
    # they all hold, only computes this step's coefficients (the same Python doubles as prepare_step) and launches; anything
    # else -- a gradient somewhere else, another set of tensors with a gradient, changed betas, a different early offer -- falls
    # back to the general path, which re-plans.

    def _make_plan(self, offer_addresses, early_addresses):
        import ctypes
        recs, groups = [], []
        for group in self.param_groups:
            groups.append((group, group["params"], len(group["params"]),
                           (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))))
            for p in group["params"]:
                g = p.grad
                if g is None:
                    recs.append((p, None, None, None, None, None, None))
                    continue
                st = self.state[p]
                if g.dtype != torch.float32 or not self._layout_ok(g, p) or not p.is_cuda:
                    self._plan = None      # (a gradient that needs a layout copy every step: the general path does that)
                    return
                recs.append((p, st, int(p.data_ptr()), int(g.data_ptr()), g.stride(), st["exp_avg"], st["exp_avg_sq"]))
        live = [r for r in recs if r[1] is not None]
        if not live or len({g[3] for g in groups}) != 1:
            self._plan = None
            return

        def items(sel):
            arr = (JtAdamItem * max(len(sel), 1))()
            for k, j in enumerate(sel):
                p, st, pp, gp, _, m, v = live[j]
                arr[k].p, arr[k].g, arr[k].m, arr[k].v, arr[k].n = pp, gp, ptr(m), ptr(v), p.numel()
            return arr, (ctypes.c_float * max(2 * len(sel), 2))(), sel
        every = list(range(len(live)))
        if early_addresses is None:
            parts = (items(every),)
        else:
            parts = (items([j for j in every if live[j][3] in early_addresses]),
                     items([j for j in every if live[j][3] not in early_addresses]))
            if not parts[0][2]:
                self._plan = None
                return
        self._plan = dict(recs=recs, groups=groups, parts=parts, offer=offer_addresses, key=groups[0][3], dev=live[0][0].device)

    def _planned_step(self, early_ok):
        plan = self.__dict__.get("_plan")
        if plan is None:
            return False
        from . import ops
        groups = plan["groups"]
        if len(self.param_groups) != len(groups):
            return False
        for have, (group, plist, n, key) in zip(self.param_groups, groups):
            if have is not group or group["params"] is not plist or len(plist) != n or \
                    (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"])) != key:
                return False
        offer = None
        if ops._EARLY_GRADS:
            if not early_ok:
                ops._EARLY_GRADS.clear()
            else:
                offer = ops._EARLY_GRADS.get(ops.device_key(plan["dev"]))
        if (None if offer is None else offer[1]) != plan["offer"]:
            return False
        for p, st, pp, gp, gs, m, v in plan["recs"]:
            g = p.grad
            if st is None:
                if g is not None:
                    return False
                continue
            # (the gradient's dtype is the parameter's -- torch refuses any other .grad --; the parameter's state entry is replaced
            #  only by load_state_dict, which drops the plan)
            if (g is None or g.data_ptr() != gp or p.data_ptr() != pp or g.stride() != gs
                    or st["exp_avg"] is not m or st["exp_avg_sq"] is not v):
                return False
        # everything is where the plan was made: this step's coefficients, in prepare_step's arithmetic
        b1, b2, _ = plan["key"]
        c1, c2 = {}, {}
        coefs = []
        gi = iter(groups)
        left = 0
        for p, st, pp, gp, gs, m, v in plan["recs"]:
            while left == 0:
                group, _, left, _ = next(gi)
                lr = float(group["lr"])
            left -= 1
            if st is None:
                continue
            t = float(st["step"]) + 1.0
            st["step"] = t
            a = c1.get(t)
            if a is None:
                a, b = c1[t], c2[t] = 1.0 - b1 ** t, 1.0 / math.sqrt(1.0 - b2 ** t)
            else:
                b = c2[t]
            coefs.append(lr / a)
            coefs.append(b)
        eps = plan["key"][2]
        parts = plan["parts"]

        def launch(part, stream):
            arr, host, sel = part
            if len(sel) == len(coefs) // 2:
                host[:] = coefs
            else:
                host[:] = [c for j in sel for c in (coefs[2 * j], coefs[2 * j + 1])]
            check(lib.jt_adam_step_coefs(arr, len(sel), b1, b2, eps, host, stream), "jt_adam_step_coefs")
        self.planned_steps = getattr(self, "planned_steps", 0) + 1
        if len(parts) == 1:
            launch(parts[0], _stream())
            return True
        ev, _, storage, aux = ops.take_early_grads(plan["dev"])
        aux.wait_event(ev)
        storage.record_stream(aux)
        launch(parts[0], aux.cuda_stream)
        done = self.__dict__.setdefault("_early_done", {}).setdefault(ops.device_key(plan["dev"]), torch.cuda.Event())
        done.record(aux)
        if parts[1][2]:
            launch(parts[1], _stream())
        torch.cuda.current_stream().wait_event(done)
        self.early_steps = getattr(self, "early_steps", 0) + 1
        return True
