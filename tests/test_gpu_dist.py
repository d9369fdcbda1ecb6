"""Ray-sharded data parallelism through the path a multi-GPU run takes: two FRESH processes (gloo
rendezvous, both on GPU 0), Model.enable_data_parallel -> the renderer's backward all-reduces its flat gradient buffer in
three collectives, pose gradients through Model.reduce_pose_gradients.  Checked per sharding mode: (1) the reduced
gradient of the first step equals the single-process gradient -- "pixel" / "view": of THE iteration one process renders
(same draws, all views, whole lattice: the ranks' shards partition it); "offset": of the union of the two ranks' own
lattices; (2) after three optimizer steps both ranks hold identical parameters.  Last: `bench.py --gpus 2` with no
launcher around it starts its own two ranks."""
import json
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


@pytest.mark.parametrize("mode", ["pixel", "view", "offset"])
def test_two_ranks_equal_one_process(tmp_path, mode):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "3", mode],
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

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as W
    from joint_tensorf_amd.options import Opt
    opt, model, var = W.build()
    g = model.graph
    tf = g.nerf.tensorf
    model.optim.zero_grad()
    model.optim_pose.zero_grad()
    if mode == "offset":
        # (1) one process, the two lattices one after the other (gradients accumulate), same draws and jitter
        l1 = opt.loss_weight.L1
        for rank in range(2):
            np.random.seed(100)
            tf.jitter_override = W.jitter_for(rank)
            g.ray_shard = ("offset", rank, 2)
            g.it = model.it
            if rank == 1:  # the regularisers are not part of the exchange: every rank adds them once
                opt.loss_weight.L1 = Opt(init=0.0, rest=0.0)
            v = g.forward(opt, Opt(dict(var)), mode="train")
            loss = g.compute_loss(opt, v, mode="train")
            loss = model.summarize_loss(opt, v, loss)
            loss.all.backward()
        opt.loss_weight.L1 = l1
    else:
        # (1) THE iteration: one process, all four views, the whole lattice, the same draws and per-ray jitter
        assert r0["first"]["rays"] + r1["first"]["rays"] > 0
        ids, n_all = W.global_ray_ids(opt, g, 4, 100, ("none", 0, 1))
        assert n_all == r0["first"]["rays"] + r1["first"]["rays"], "the two shards partition the iteration's rays"
        tf.jitter_override = W.jitter_for(0, n_all)
        g.it = model.it
        v = g.forward(opt, Opt(dict(var)), mode="train")
        assert v.rgb.shape[0] * v.rgb.shape[1] == n_all
        loss = g.compute_loss(opt, v, mode="train")
        loss = model.summarize_loss(opt, v, loss)
        loss.all.backward()
    worst = ("", 0.0)
    for k, p in g.named_parameters():
        if p.grad is None:
            continue
        a, b = r0["first"][k].double(), p.grad.detach().cpu().double()
        e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        if e > worst[1]:
            worst = (k, e)
        assert e <= 1e-5, (k, e)
    print("%s shards, two ranks vs one process: worst relative gradient difference %.2e (%s); %d + %d rays"
          % (mode, worst[1], worst[0], r0["first"]["rays"], r1["first"]["rays"]))


def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher: the script spawns two fresh ranks before touching the GPU (here both
    on GPU 0 over gloo -- the one-GPU functional form), shards ONE iteration over them and prints n_gpus = 2."""
    env = dict(os.environ, JT_BENCH_SINGLE_DEVICE="1", JT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stage", "0", "--total-rays", "8192",
                        "--steps", "3", "--warmup", "1", "--no-probe"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    rk = line["ranks"]
    assert rk["world_size"] == 2 and rk["backend"] == "gloo" and rk["rccl_ranks"] == 0 and rk["shard"] == "pixel"
    assert rk["launcher"].startswith("bench.py --gpus N")
    per = rk["rays_per_iter_per_rank"]
    # 8192 nominal rays over 100 views: 81 per view -> stride ceil(sqrt(160000 // 81)) = 45 -> 8 or 9 points per axis
    assert len(per) == 2 and abs(per[0] - per[1]) <= 100 and 6400 <= per[0] + per[1] <= 8100
    assert "allreduce_ms" in line and len(line["allreduce_ms"]) == 3
    # the record says which single-GPU figure it scales from, and how much of the exchange the backward did not hide
    n1 = line["strong_scaling_n1"]
    assert "--gpus 1 --total-rays 8192" in n1["workload"] and "configs3_single_gpu" in n1["key"]
    assert line["ms_per_step_no_collectives"] > 0
    assert abs(line["allreduce_overlap_ms"] - (line["ms_per_step"] - line["ms_per_step_no_collectives"])) < 1e-6


@pytest.mark.parametrize("shard", ["pixel", "view"])
def test_bench_gpus8_single_device(shard):
    """`python bench.py --gpus 8` functionally: eight fresh ranks, all on GPU 0 over gloo, share ONE
    configs[3] iteration (65 536 nominal = 62 500 lattice rays: 25 x 25 points on each of 100 views).  pixel shards: every rank
    takes 78 or 79 of a view's 625 lattice points (7 800 / 7 900 rays); view shards: 12 or 13 whole views (7 500 / 8 125 rays).
    After the timed steps every rank holds the SAME parameters (replicated optimizer step on all-reduced gradients), and the
    line names the single-GPU figure it scales from.  No scaling number is asserted: eight ranks on one device say nothing
    about xGMI."""
    env = dict(os.environ, JT_BENCH_SINGLE_DEVICE="1", JT_DIST_BACKEND="gloo", JT_BENCH_CHECKSUM="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--shard", shard, "--no-probe", "--no-cpu-baseline", "--no-torch-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong"
    rk = line["ranks"]
    assert rk["world_size"] == 8 and rk["backend"] == "gloo" and rk["rccl_ranks"] == 0 and rk["shard"] == shard
    per = [int(round(v)) for v in rk["rays_per_iter_per_rank"]]
    assert len(per) == 8 and sum(per) == 62500, per
    if shard == "pixel":
        assert sorted(set(per)) == [7800, 7900] and per.count(7900) == 625 - 8 * 78, per   # 79 / 78 points x 100 views
    else:
        assert sorted(set(per)) == [7500, 8125] and per.count(8125) == 100 - 8 * 12, per   # 13 / 12 views x 625 points
    pc = line["param_checksum"]
    assert pc["ranks_compared"] == 8 and pc["identical_on_every_rank"] is True, pc
    assert all(pc[k] > 0 for k in ("density", "app", "mlp")), pc
    assert "strong_scaling_n1" in line and "--gpus 1 --total-rays 65536" in line["strong_scaling_n1"]["workload"]
    assert len(line["allreduce_ms"]) == 3


def test_bench_gpus_mismatch_is_an_error():
    """--gpus 4 inside a process group of one rank: refuse instead of printing a line for another N"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and "--gpus 4" in r.stderr


def test_rccl_process_group_of_one_rank():
    """The collectives of the N > 1 path ON RCCL (every other test in this file rendezvous over gloo): a
    process group of one rank on the "nccl" backend (= RCCL on ROCm), `init_process_group("nccl", device_id=...)`, the three
    asynchronous all-reduces on slices of the flat gradient buffer inside the renderer's backward with their record_stream
    bookkeeping, the pose all-reduce behind it, the barrier + max-over-ranks timing of the bench line.  One rank is all a
    one-GPU box allows; what it proves is that this code path initialises and runs on RCCL at all before the driver's 8-GPU
    node is the first to try."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               JT_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("JT_DIST_BACKEND", None)
    env.pop("JT_BENCH_SINGLE_DEVICE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-probe", "--no-cpu-baseline", "--no-torch-baseline", "--no-extras", "--no-live-pmc"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    rk = line["ranks"]
    assert rk["backend"] == "nccl" and rk["rccl_ranks"] == 1 and rk["world_size"] == 1, rk
    assert rk["devices"] == "one GPU per rank"
    assert len(line["allreduce_ms"]) == 3 and all(v > 0 for v in line["allreduce_ms"].values()), line["allreduce_ms"]
    assert np.isfinite(line["value"]) and line["value"] > 0 and line["ms_per_step"] > 0
    assert line["ms_per_step_no_collectives"] > 0
