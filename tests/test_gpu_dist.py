"""Ray-sharded data parallelism through the path a multi-GPU run takes: two FRESH processes (gloo
rendezvous, both on GPU 0), ops.set_data_parallel -> the renderer's backward all-reduces its flat gradient buffer in
three collectives, pose gradients through dist.allreduce_gradients.  Checked: (1) the reduced gradient of the first step
equals the single-process gradient of the UNION of the two ranks' lattices; (2) after three optimizer steps both ranks
hold identical parameters."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_one_process_on_the_union_of_their_rays(tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "3"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    # (2) replicated optimizer steps on identical reduced gradients: bit-identical parameters on both ranks
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    for k in r0["first"]:
        if k != "rays":
            assert torch.equal(r0["first"][k], r1["first"][k]), k
    assert r0["first"]["rays"] == r1["first"]["rays"]

    # (1) one process, the two lattices one after the other (gradients accumulate), same draws and jitter
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as W
    from joint_tensorf_amd.options import Opt
    opt, model, var = W.build()
    g = model.graph
    tf = g.nerf.tensorf
    model.render_loss_scale = 0.5
    model.optim.zero_grad()
    model.optim_pose.zero_grad()
    l1 = opt.loss_weight.L1
    for rank in range(2):
        np.random.seed(100)
        tf.jitter_override = W.jitter_for(rank)
        g.lattice_rank = (rank, 2)
        g.it = model.it
        if rank == 1:  # the regularisers are not part of the exchange: every rank adds them once
            opt.loss_weight.L1 = Opt(init=0.0, rest=0.0)
        v = g.forward(opt, Opt(dict(var)), mode="train")
        loss = g.compute_loss(opt, v, mode="train")
        loss = model.summarize_loss(opt, v, loss)
        loss.all.backward()
    opt.loss_weight.L1 = l1
    worst = ("", 0.0)
    for k, p in g.named_parameters():
        if p.grad is None:
            continue
        a, b = r0["first"][k].double(), p.grad.detach().cpu().double()
        e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        if e > worst[1]:
            worst = (k, e)
        assert e <= 1e-5, (k, e)
    print("two ranks vs the union in one process: worst relative gradient difference %.2e (%s); %d rays per rank"
          % (worst[1], worst[0], r0["first"]["rays"]))
