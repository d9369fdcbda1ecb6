#!/usr/bin/env python3
"""One rank of the two-process data-parallel test (tests/test_gpu_dist.py): a FRESH process, gloo rendezvous, the
default overlap path (ops.set_data_parallel: the renderer's backward all-reduces its own gradients), every rank on
GPU 0.  Writes the first step's gradients and the parameters after `steps` steps to <out>/rank<r>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def build(opt_over=None):
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views
    torch.manual_seed(0)
    np.random.seed(0)
    opt = make_options("bat_blender_VM", device="cuda", data=dict(image_size=[64, 64], num_views=4),
                       train_schedule=dict(n_voxel_init=24 ** 3, n_rays_init=256, n_rays_rest=256, upsample_iters=[10 ** 9]),
                       nerf=dict(n_rays=256), c2f_mode="None")
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=4)
    model.setup_optimizer(opt)
    with torch.no_grad():
        for p in model.graph.nerf.tensorf.density_plane:
            p.mul_(22.0)
        model.graph.se3_refine.weight.copy_(0.01 * torch.randn(4, 6, device="cuda"))
    var = make_views(opt, 4, seed=3, device="cuda")
    return opt, model, var


def jitter_for(rank, n=8192):
    return torch.rand(n, 1, generator=torch.Generator().manual_seed(50 + rank)).cuda()


def global_ray_ids(opt, graph, n_views, seed, shard):
    """The rays of the iteration whose host draws start at np.random.seed(seed) that `shard` = (mode, rank, world)
    renders, as indices into the single-process iteration's [view, lattice point] ray list (view-major), plus that
    list's length; the generator is left where the iteration expects it (re-seeded)."""
    from joint_tensorf_amd.dist import shard_indices
    np.random.seed(seed)
    step = graph.lattice_step(opt, n_views)
    ox, oy = np.random.randint(step), np.random.randint(step)
    n_pts = len(range(ox, opt.W, step)) * len(range(oy, opt.H, step))
    mode, rank, world = shard
    views = shard_indices(n_views, rank, world) if mode == "view" else list(range(n_views))
    pts = shard_indices(n_pts, rank, world) if mode == "pixel" else list(range(n_pts))
    np.random.seed(seed)
    return torch.tensor([v * n_pts + p for v in views for p in pts]), n_views * n_pts


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else "offset"
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joint_tensorf_amd.options import Opt
    opt, model, var = build()
    model.enable_data_parallel(opt, rank, world, shard=mode)
    var = model.local_views(var)
    tf = model.graph.nerf.tensorf
    first = None
    for it in range(steps):
        np.random.seed(100 + it)          # the host draws are shared by all ranks
        g = model.graph
        if mode == "offset":              # a lattice of its own per rank: a jitter stream per rank
            tf.jitter_override = jitter_for(rank)
        else:                             # a shard of THE iteration: the jitter of the rays it renders
            ids, n_all = global_ray_ids(opt, g, 4, 100 + it, (mode, rank, world))
            tf.jitter_override = jitter_for(0, n_all)[ids.cuda()]
        g.it = model.it
        model.optim.zero_grad()
        model.optim_pose.zero_grad()
        v = g.forward(opt, Opt(dict(var)), mode="train")
        loss = g.compute_loss(opt, v, mode="train")
        loss = model.summarize_loss(opt, v, loss)
        loss.all.backward()
        model.reduce_pose_gradients()     # what left through the rays
        if first is None:
            first = {k: p.grad.detach().cpu().clone() for k, p in g.named_parameters() if p.grad is not None}
            first["rays"] = v.rgb.shape[0] * v.rgb.shape[1]
        model.optim.step()
        model.optim_pose.step()
        model.it += 1
        g.nerf.set_progress(model.it / opt.max_iter)
        model.after_iteration(opt)
    params = {k: p.detach().cpu().clone() for k, p in model.graph.named_parameters()}
    torch.save(dict(first=first, params=params), os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
