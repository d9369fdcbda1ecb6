"""Checkpoints in the reference's format (util.py:120-184): one torch-pickled dict
    {epoch, iter, graph: graph.state_dict(), opt, manually_tracked_parameters: model.save_param_state(),
     optim*/sched*: their state_dicts}
written to <output_path>/model.ckpt (+ <output_path>/model/<iter>.ckpt).  State-dict key names and logical shapes are
the reference's, so files interchange both ways; factor tensors load by value into the
channel-last storage of this build."""
import os
import shutil
import sys
import types

import torch

from .options import Opt


def get_child_state_dict(state_dict, key):
    return {".".join(k.split(".")[1:]): v for k, v in state_dict.items() if k.startswith("{}.".format(key))}


def _plain(x):
    """Opt -> dict, recursively (what gets pickled must not need this package to be read back)"""
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    return x


def save_checkpoint(opt, model, ep, it, latest=False, children=None):
    os.makedirs("{0}/model".format(opt.output_path), exist_ok=True)
    sd = model.graph.state_dict()
    if children is not None:
        sd = {k: v for k, v in sd.items() if k.startswith(children)}
    # contiguous NCHW copies: a reader without this package (the reference) sees ordinary tensors
    sd = {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in sd.items()}
    ckpt = dict(epoch=ep, iter=it, graph=sd, opt=_plain(opt))
    if hasattr(model, "save_param_state"):
        ckpt["manually_tracked_parameters"] = model.save_param_state()
    for key in model.__dict__:
        if key.split("_")[0] in ("optim", "sched") and getattr(model, key) is not None:
            ckpt[key] = getattr(model, key).state_dict()
    path = "{0}/model.ckpt".format(opt.output_path)
    torch.save(ckpt, path)
    if not latest:
        shutil.copy(path, "{0}/model/{1}.ckpt".format(opt.output_path, ep or it))
    return path


def _load(path, device):
    # a checkpoint written by the reference pickles its `opt` as easydict.EasyDict; that package may be absent here
    if "easydict" not in sys.modules:
        try:
            import easydict  # noqa: F401
        except ImportError:
            stub = types.ModuleType("easydict")
            stub.EasyDict = Opt
            sys.modules["easydict"] = stub
    return torch.load(path, map_location=device, weights_only=False)


def restore_checkpoint(opt, model, load_name=None, resume=False):
    """util.restore_checkpoint (util.py:120-160): manually tracked state first (it resizes the scene tensors and rebuilds
    the optimizer), then the graph's state_dict (non-strict), then -- only when resuming -- optimizer / scheduler state."""
    assert (load_name is None) == (resume is not False)
    if resume:
        load_name = "{0}/model.ckpt".format(opt.output_path) if resume is True else \
            "{0}/model/{1}.ckpt".format(opt.output_path, resume)
    ckpt = _load(load_name, opt.device)
    if hasattr(model, "load_param_state") and "manually_tracked_parameters" in ckpt:
        ps = dict(ckpt["manually_tracked_parameters"])
        ps["iter"] = ckpt["iter"]
        model.load_param_state(opt, ps)
    graph_sd = dict(ckpt["graph"])
    n_train = len(model.train_data) if getattr(model, "train_data", None) is not None else None
    if "se3_refine.weight" in graph_sd and n_train is not None and graph_sd["se3_refine.weight"].shape[0] != n_train:
        del graph_sd["se3_refine.weight"]  # another train split: the pose embedding does not carry over
    for name, child in model.graph.named_children():
        child_sd = get_child_state_dict(graph_sd, name)
        if child_sd:
            child.load_state_dict(child_sd, strict=False)
    model.graph.load_state_dict(graph_sd, strict=False)
    nerf = getattr(model.graph, "nerf", None)
    if nerf is not None and hasattr(nerf, "progress_host"):  # host mirrors of the checkpointed progress Parameters
        nerf.progress_host = float(nerf.progress.detach().cpu())
        nerf.test_time_progress_host = float(nerf.test_time_progress.detach().cpu())
    for key in list(model.__dict__):
        if key.split("_")[0] in ("optim", "sched") and key in ckpt and resume and getattr(model, key) is not None:
            getattr(model, key).load_state_dict(ckpt[key])
    if resume:
        ep, it = ckpt["epoch"], ckpt["iter"]
        if resume is not True:
            assert resume == (ep or it)
    else:
        ep, it = None, None
    return ep, it
