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
