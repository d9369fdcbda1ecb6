"""Options: attribute dict + yaml loading with `_parent_` inheritance.

Same semantics as the reference's options.load_options / override_options (options.py:72-108):
a child yaml overrides its parent key by key, recursively.  The engine only *reads* `opt`; an
EasyDict built by the reference's own options.py works just as well.
"""
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


class Opt(dict):
    """dict with attribute access, nested dicts converted recursively."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(e) for e in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


def _override(base, over):
    for k, v in over.items():
        if isinstance(v, dict):
            base[k] = _override(base.get(k, Opt()) if isinstance(base.get(k), dict) else Opt(), v)
        else:
            base[k] = v
    return base


def load_options(name_or_path):
    path = name_or_path
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(CONFIG_DIR, name_or_path if name_or_path.endswith(".yaml") else name_or_path + ".yaml")
    with open(path) as f:
        opt = Opt(yaml.safe_load(f))
    if "_parent_" in opt:
        parent = opt.pop("_parent_")
        base = load_options(os.path.join(os.path.dirname(path), parent))
        opt = _override(base, opt)
    return opt


def make_options(name, device="cuda", **overrides):
    """Load a config and fill the run-time fields the engine expects (opt.device, opt.H, opt.W)."""
    opt = load_options(name)
    _override(opt, Opt(overrides))
    opt.device = device
    opt.H, opt.W = (int(v) for v in opt.data.image_size)
    return opt


def compress_schedule(opt, factor):
    """Shorten a run by `factor` without changing its SHAPE: every iteration-denominated key of the yaml (max_iter, grid
    upsamplings, alpha-mask updates, ray-count / gradient-accumulation switches, pose reset and warm-up, edge-loss and
    TV_depth horizons) is divided by it; the schedules that are functions of progress = it / max_iter (factor blur, 2-D
    blur, near plane, learning-rate decay) follow by themselves.  Used by the convergence acceptance test and
    tools/converge.py: the same stages in the same order, fewer iterations in each."""
    f = float(factor)
    if f == 1.0:
        return opt

    def div(v):
        return max(1, int(round(v / f)))
    opt.max_iter = div(opt.max_iter)
    ts = opt.train_schedule
    for k in ("upsample_iters", "update_alphamask_iters", "reset_pose_on_iters"):
        if k in ts and ts[k] is not None:
            ts[k] = [div(v) for v in ts[k]]
    for k in ("change_n_rays_after_n_iters", "change_n_AccumPoseGrad_after_n_iters", "change_n_AccumGrad_after_n_iters",
              "reset_pose_on_iter", "all_view_sample_after_n_iters", "single_view_sample_after_n_iters"):
        if k in ts and ts[k] is not None:
            ts[k] = div(ts[k])
    if "edge_mask_before_iter" in opt and opt.edge_mask_before_iter is not None:
        opt.edge_mask_before_iter = div(opt.edge_mask_before_iter)
    if opt.optim.get("warmup_pose", None):
        opt.optim.warmup_pose = div(opt.optim.warmup_pose)
    if "TV_depth_until_iters" in opt.loss_weight and opt.loss_weight.TV_depth_until_iters is not None:
        opt.loss_weight.TV_depth_until_iters = div(opt.loss_weight.TV_depth_until_iters)
    return opt
