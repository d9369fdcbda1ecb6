"""world_size-2 gloo test of the gradient exchange (the N>1 path of bench.py), on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("jt_dist", os.path.join(here, "joint_tensorf_amd", "dist.py"))
    jd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(jd)  # dist.py alone: no HIP library needed for the exchange logic
    torch.manual_seed(0)
    # a channel-last "plane" (big), a line + MLP weights (small), one parameter without grad
    plane = torch.nn.Parameter(torch.zeros(600, 600, 16).permute(2, 0, 1)[None])
    line = torch.nn.Parameter(torch.zeros(1, 16, 9, 1))
    w = torch.nn.Parameter(torch.zeros(64, 150))
    nog = torch.nn.Parameter(torch.zeros(3))
    g = torch.Generator().manual_seed(100 + rank)
    plane.grad = torch.randn(600, 600, 16, generator=g).permute(2, 0, 1)[None]
    line.grad = torch.randn(1, 16, 9, 1, generator=g)
    w.grad = torch.randn(64, 150, generator=g)
    assert not plane.grad.is_contiguous()
    jd.allreduce_gradients([plane, line, w, nog], world)
    # expected: sum over ranks of the same generators
    exp_p = exp_l = exp_w = 0
    for r in range(world):
        gg = torch.Generator().manual_seed(100 + r)
        exp_p = exp_p + torch.randn(600, 600, 16, generator=gg).permute(2, 0, 1)[None]
        exp_l = exp_l + torch.randn(1, 16, 9, 1, generator=gg)
        exp_w = exp_w + torch.randn(64, 150, generator=gg)
    ok = torch.allclose(plane.grad, exp_p) and torch.allclose(line.grad, exp_l) and torch.allclose(w.grad, exp_w) \
        and nog.grad is None
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_allreduce_gradients_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _worker_flat(rank, world, port, ret):
    """The exchange the renderer's backward really issues (ops.DpReducer over the one flat gradient buffer of
    ops._zeros_flat: three collectives -- appearance factors, density factors, basis + MLP -- then the regulariser
    gradient added on every rank), with a CPU stand-in for the kernels that fill the buffer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joint_tensorf_amd import ops
    # the tensors of a (small, non-cubic) scene in storage layout: planes [H,W,C], lines [L,C], basis + MLP; odd sizes
    # so that the 16-byte rounding of the offsets inside the flat buffer is exercised
    dp = [torch.zeros(7, 9, 16), torch.zeros(5, 9, 16), torch.zeros(5, 7, 16)]
    dl = [torch.zeros(5, 16), torch.zeros(7, 16), torch.zeros(9, 16)]
    ap = [torch.zeros(7, 9, 20), torch.zeros(5, 9, 20), torch.zeros(5, 7, 20)]
    al = [torch.zeros(5, 20), torch.zeros(7, 20), torch.zeros(9, 20)]
    mlp = [torch.zeros(20, 60), torch.zeros(32, 100), torch.zeros(32), torch.zeros(32, 32), torch.zeros(32),
           torch.zeros(3, 44), torch.zeros(3)]
    groups = [dp, dl, ap, al, mlp]
    views, gflat, spans = ops._zeros_flat(groups, with_flat=True)
    assert len(spans) == 5 and spans[0][0] == 0 and all(spans[k][1] == spans[k + 1][0] for k in range(4))
    assert all(o % 4 == 0 for s_ in spans for o in s_) and spans[4][1] == gflat.numel()

    def fill(r):  # what rank r's render backward leaves in the views
        g = torch.Generator().manual_seed(7 + r)
        return [[torch.randn(t.shape, generator=g) for t in grp] for grp in groups]
    mine = fill(rank)
    for vg, mg in zip(views, mine):
        for v, m_ in zip(vg, mg):
            v.copy_(m_)
    red = ops.DpReducer(gflat, spans)
    red.reduce(2, 3)   # appearance planes + lines
    red.reduce(0, 1)   # density planes + lines
    red.reduce(4, 4)   # basis + MLP
    red.wait()
    reg = [torch.full(t.shape, 0.25) for t in dp]  # the regulariser gradient: same on every rank, added AFTER the exchange
    for v, r_ in zip(views[0], reg):
        v.add_(r_)
    every = [fill(r) for r in range(world)]
    ok = True
    for gi, vg in enumerate(views):
        for ti, v in enumerate(vg):
            exp = sum(every[r][gi][ti] for r in range(world))
            if gi == 0:
                exp = exp + reg[ti]
            ok = ok and torch.allclose(v, exp, atol=1e-6)
    # padding floats between tensors stay zero (they are part of the collectives)
    used = torch.zeros(gflat.numel(), dtype=torch.bool)
    for vg in views:
        for v in vg:
            off = (v.data_ptr() - gflat.data_ptr()) // 4
            used[off:off + v.numel()] = True
    ok = ok and bool((gflat[~used] == 0).all())
    ret[rank] = bool(ok) and sorted(ops._DP["span_elems"].values()) == sorted(
        [spans[3][1] - spans[2][0], spans[1][1] - spans[0][0], spans[4][1] - spans[4][0]])
    dist.destroy_process_group()


def test_flat_buffer_exchange_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_flat, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_rank_lattice_offsets_keep_the_ray_count_and_differ():
    """dist.rank_lattice_offset: every rank renders the same number of lattice points as the shared draw would, on
    other pixels (the rank-sharding statement of SURVEY 8(e))."""
    from joint_tensorf_amd.dist import rank_lattice_offset
    for step, extent in ((90, 400), (45, 400), (16, 400), (17, 640)):
        for o in range(step):
            n = len(range(o, extent, step))
            offs = [rank_lattice_offset(o, step, extent, r, 8) for r in range(8)]
            assert all(len(range(x, extent, step)) == n for x in offs) and offs[0] == o
            cls = [x for x in range(step) if len(range(x, extent, step)) == n]
            assert len(set(offs)) == min(8, len(cls))


def test_two_axis_offsets_give_every_rank_its_own_pixels():
    """dist.rank_lattice_offsets (what Graph.forward uses under "offset" sharding): the shared draw's lattice SHAPE on
    every rank, and -- unlike the per-axis shift, which repeats as soon as one class has fewer members than ranks (the
    5-member class of stride 45 on 400 pixels) -- pairwise different (ox, oy) while |class_x| * |class_y| >= world."""
    from joint_tensorf_amd.dist import _offset_class, rank_lattice_offsets
    for step, W, H in ((90, 400, 400), (45, 400, 400), (16, 400, 400), (17, 640, 480), (37, 640, 480)):
        for ox in range(0, step, 3):
            for oy in range(0, step, 5):
                shape = (len(range(ox, W, step)), len(range(oy, H, step)))
                offs = [rank_lattice_offsets(ox, oy, step, W, H, r, 8) for r in range(8)]
                assert offs[0] == (ox, oy)
                assert all((len(range(x, W, step)), len(range(y, H, step))) == shape for x, y in offs)
                if len(_offset_class(ox, step, W)) * len(_offset_class(oy, step, H)) >= 8:
                    assert len(set(offs)) == 8, (step, ox, oy, offs)


def test_shards_partition_the_iteration():
    """dist.shard_indices: the interleaved split behind the "pixel" and "view" shardings -- over all ranks the shards are
    a partition, sizes differ by at most one (62 500 rays of configs[3]: 625 lattice points or 100 views over 8 ranks)."""
    from joint_tensorf_amd.dist import shard_indices
    for n, world in ((625, 8), (100, 8), (100, 2), (100, 4), (7, 8), (0, 2)):
        parts = [shard_indices(n, r, world) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
