"""Data-parallel gradient exchange: one process per GPU, ray batches sharded across ranks, parameter
gradients summed with RCCL all-reduce over xGMI.  The reference has no collective
(options.py:126 asserts a single GPU); this is the build's own scheme.

Large tensors (VM planes, up to 30 MB each at 400^3) are reduced in place through a contiguous view
of their channel-last storage -- no staging copy; the many small tensors (lines, basis, MLP, se3)
are packed into one flat bucket so the exchange is a handful of collectives per iteration.
"""
import torch
import torch.distributed as dist

SMALL_BYTES = 1 << 20


def _offset_class(o, step, extent):
    n = len(range(o, extent, step))
    return [x for x in range(step) if len(range(x, extent, step)) == n]


def rank_lattice_offset(o, step, extent, rank, world):
    """One axis of the all_view_rand_grid lattice (model/nerf.py:655-673) for rank `rank` of `world` ("offset"
    sharding, the weak-scaling mode: every rank renders a lattice of its own).

    `o` in [0, step) is the draw SHARED by all ranks (same seeded host generator everywhere).
    The number of lattice points along the axis, len(range(o, extent, step)), takes two values depending on
    which side of a threshold `o` falls; a rank that drew its own offset would therefore render 1600 / 2000 /
    2500 rays at random and every iteration of the job would run at the pace of the rank with the most.  Each
    rank instead shifts the shared draw cyclically INSIDE the shared draw's class: different pixels per
    rank, the same ray count on every rank.  (Per axis: classes with fewer members than ranks repeat; the
    two-axis form below is what Graph.forward uses.)"""
    cls = _offset_class(o, step, extent)
    k = cls.index(o)
    return cls[(k + (rank * len(cls)) // max(world, 1)) % len(cls)]


def rank_lattice_offsets(ox, oy, step, width, height, rank, world):
    """Both axes at once: the ranks are spread over the |class_x| x |class_y| offsets that give the shared draw's
    lattice shape, so that no two ranks render the same pixels while that product is >= world (a per-axis shift
    repeats as soon as ONE class has fewer members than ranks, e.g. the 5-member class of step 45 on 400 pixels)."""
    cx, cy = _offset_class(ox, step, width), _offset_class(oy, step, height)
    kx, ky = cx.index(ox), cy.index(oy)
    n = len(cx) * len(cy)
    k = (rank * n) // max(world, 1)      # distinct for distinct ranks while n >= world
    return cx[(kx + k % len(cx)) % len(cx)], cy[(ky + k // len(cx)) % len(cy)]


def shard_indices(n, rank, world):
    """The members rank::world of range(n): the interleaved split used for the exact-iteration shardings ("view":
    n = views of the training set, "pixel": n = points of the iteration's pixel lattice).  Over all ranks the
    shards partition range(n); their sizes differ by at most one."""
    return list(range(rank, n, world))


def _contiguous_view(g):
    """A contiguous alias of g's memory (channel-last factors are contiguous after a permute)."""
    if g.is_contiguous():
        return g
    if g.dim() == 4 and g.permute(0, 2, 3, 1).is_contiguous():
        return g.permute(0, 2, 3, 1)
    return None


def allreduce_gradients(params, world, group=None, force=False):
    """SUM-all-reduce the .grad of every parameter that has one (callers scale the loss by 1/world)."""
    if world == 1 and not force:
        return
    big, small = [], []
    for p in params:
        g = p.grad
        if g is None:
            continue
        v = _contiguous_view(g)
        if v is not None and v.numel() * v.element_size() >= SMALL_BYTES:
            big.append(v)
        else:
            small.append(g)
    works = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group, async_op=True) for v in big]
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g in small:
            n = g.numel()
            g.copy_(flat[off:off + n].view(g.shape))
            off += n
    for w in works:
        w.wait()
