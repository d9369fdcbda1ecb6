#!/usr/bin/env python3
"""bench.py -- train rays/s of the joint pose + TensoRF-VM training step on MI355X.

A "step" is one full optimisation iteration of the reference's hot loop on a synthetic
Blender-lego-shaped scene: pose composition from se(3) parameters -> rays for the
lattice pixels -> sampling / VM interpolation / MLP / compositing -> edge-weighted MSE + L1 (+TV)
-> backward to the VM factors, basis, MLP and se(3) -> Adam steps (scene + pose) -> LR schedule.
Inputs (images, cameras, parameters) are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config bat_blender_VM] [--stage 4]

N = 1 (default): BASELINE.json configs[1], the bat_blender_VM final-stage training step (about 2 000 rays per iteration).
N > 1: one rank per GPU over RCCL.  Under a launcher (torch.distributed.run: RANK / WORLD_SIZE / MASTER_* in the
environment) this process is one rank; without one, `python bench.py --gpus N` starts its own N rank processes before it
touches a GPU and relays rank 0's line.  The workload is BASELINE.json configs[3]: ONE 65 536-nominal-ray iteration
(step-16 lattice over 100 views = 62 500 rays, model/nerf.py:655-673) split over the ranks -- STRONG scaling; the ranks
together render exactly the rays one process would (`--shard pixel`: every rank takes the lattice points rank::N of all
views; `--shard view`: the views rank::N on the whole lattice), the VM-factor / basis / MLP gradients are summed by three
all-reduces inside the backward and the pose gradients by one after it.  The N = 1 point of that curve is
`--gpus 1 --total-rays 65536` (also reported as extra.configs3_single_gpu of the default line).  `--weak` instead keeps
the yaml's ray count on every rank, each on a lattice of its own.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="bat_blender_VM")
    ap.add_argument("--stage", type=int, default=-1,
                    help="grid stage of the yaml's upsampling schedule (0 = initial grid, -1 = final grid)")
    ap.add_argument("--it", type=int, default=-1,
                    help="start iteration inside the chosen stage (default: stage start; the final stage starts "
                         "after the blur schedule ends, use e.g. --it 9000 for the blurred part of it)")
    ap.add_argument("--n-rays", type=int, default=0, help="override opt.nerf.n_rays (0 = yaml schedule value)")
    ap.add_argument("--n-voxel-final", type=int, default=0,
                    help="override train_schedule.n_voxel_final (e.g. 27000000 = the 300^3 of the parent yaml "
                         "options/tensorf_blender_VM.yaml that BASELINE.json's configs[1] text quotes)")
    ap.add_argument("--scene", default="random", choices=["random", "blobs", "fitted"],
                    help="random = the reference's random-init factors (every in-box sample is shaded: the worst case "
                         "and the default); blobs = a few opaque Gaussian blobs baked into the density factors "
                         "(SURVEY 8(d) structured scene: a few per cent of the samples shaded, like a trained field), random "
                         "appearance, noise images; fitted = the self-consistent scene: the model HOLDS the ground-truth "
                         "field (blobs + smooth textured appearance, synthetic.make_gt_scene resampled onto the model's grid) "
                         "and is supervised with images rendered from it at the ground-truth cameras -- the sharp last stage "
                         "of a run that has converged")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--probe-only", action="store_true",
                    help="run only the single-launch kernel probe (the target of the rocprofv3 --pmc passes)")
    ap.add_argument("--no-probe", action="store_true", help="skip the isolated-launch probe (roofline.probe)")
    ap.add_argument("--no-torch-baseline", action="store_true",
                    help="skip the stock-torch-ops GPU baseline (the oracle on device=cuda, same workload)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary workloads (stage 0 / stage 4 blurred, LLFF final grid, blob scene eager + "
                         "hipGraph), each run as a short child process after the headline")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect roofline.traffic in this run (two short rocprofv3 --pmc child passes of this command); "
                         "the committed passes under profiles/ are quoted instead")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: every rank renders the yaml's own ray count (weak scaling) instead of the default, "
                         "BASELINE.json configs[3]: 65 536 nominal rays per iteration sharded N ways (strong scaling)")
    ap.add_argument("--total-rays", type=int, default=0,
                    help="nominal rays per iteration of the strong-scaling mode (default 65536 when N > 1); given at N = 1 "
                         "it runs that whole iteration on one GPU: the N = 1 point of the strong-scaling curve")
    ap.add_argument("--shard", default=None, choices=["pixel", "view", "offset"],
                    help="how the iteration's rays are split over the ranks (Graph.ray_shard): pixel (default, strong), "
                         "view (SURVEY 8(e)), offset (every rank a lattice of its own: the --weak mode)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N FRESH rank processes -- this parent has not
    touched a GPU (torch.cuda.device_count() does not initialise one on this image) and never does -- wire them up the
    way torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay rank 0's
    stdout and leave with the worst exit code.  A rank that dies takes the others with it (by PID)."""
    import socket
    import subprocess
    n = args.gpus
    single = os.environ.get("JT_BENCH_SINGLE_DEVICE") == "1"
    ndev = torch.cuda.device_count()
    if ndev < n and not single:
        print("bench.py: --gpus %d but %d GPU(s) visible (JT_BENCH_SINGLE_DEVICE=1 JT_DIST_BACKEND=gloo runs all ranks "
              "on GPU 0 as a functional test)" % (n, ndev), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   JT_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = None
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                try:
                    if r == 0:
                        out0 = procs[0].communicate(timeout=0.5)[0]
                    else:
                        procs[r].wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if procs[r].returncode != 0:
                    rc = rc or procs[r].returncode or 1
                    for q in pending:   # the rendezvous of the others would hang: stop exactly the PIDs started here
                        procs[q].terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    return rc


# JT_FORCE_DIST=1 under `torch.distributed.run --nproc-per-node 1`: a process group of ONE rank on RCCL, all
# collectives issued -- the functional check of the N > 1 code path that a one-GPU box allows
FORCE_DIST = os.environ.get("JT_FORCE_DIST") == "1"


def setup_dist(args):
    """(world, rank, local device, backend name) of this rank process"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("JT_BENCH_SINGLE_DEVICE") == "1":
        local = 0  # functional test of the N > 1 path on a one-GPU box (with JT_DIST_BACKEND=gloo)
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        backend = os.environ.get("JT_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        world, rank = dist.get_world_size(), dist.get_rank()  # what the process group says, not what the env said
        return world, rank, local, dist.get_backend()
    torch.cuda.set_device(0)
    return world, rank, local, None


def stage_setup(opt, stage):
    """Put `opt` at the start of grid stage `stage` of its own schedule (model/tensorf.py:304,449-461)."""
    ups = list(opt.train_schedule.upsample_iters)
    n_stage = len(ups) + 1
    if stage < 0:
        stage += n_stage
    stage = max(0, min(stage, n_stage - 1))
    n_list = torch.round(torch.exp(torch.linspace(np.log(opt.train_schedule.n_voxel_init),
                                                  np.log(opt.train_schedule.n_voxel_final), n_stage))).long().tolist()
    it0 = 0 if stage == 0 else ups[stage - 1]
    # the final stage of bat_blender_VM runs 9000..40000; blur is active until progress 0.3 (it 12000).
    # Bench the sharp part of the last stage (77 % of all iterations of the run are sharp last-stage ones).
    if stage == n_stage - 1:
        it0 = max(it0, int(0.3 * opt.max_iter) + 1)
    opt.train_schedule.n_voxel_init = n_list[stage]
    opt.train_schedule.upsample_iters = [u for u in ups if u > it0] or [10 ** 9]
    return stage, it0


def build_model(opt, it0, n_views):
    from joint_tensorf_amd.model import bat_hip
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=n_views)
    model.setup_optimizer(opt)
    model.it = it0
    model.graph.nerf.set_progress(it0 / opt.max_iter)
    return model


def _cpu_oracle_rate(grid, S, rays_per_view, seconds_budget, blur_off=True):
    """fwd + loss + bwd of the reference algorithm (torch-CPU oracle port, parity-pinned) on 4 views x
    `rays_per_view` rays: median over up to 9 repetitions for each of a few intra-op thread counts (the path is
    thousands of small ops; all cores of a big host oversubscribe it badly), best thread count reported."""
    from oracle import tensorf_oracle as O
    g = torch.Generator().manual_seed(0)
    cfg = O.SceneCfg([-1.5] * 3 + [1.5] * 3, grid, [2.0, 6.0])
    params = O.init_params(grid, scale=0.1, bias=0.0, generator=g)
    for _, v in O.flat_params(params):
        v.requires_grad_(True)
    B, r, H, W = 4, rays_per_view, 400, 400
    from joint_tensorf_amd.synthetic import look_at
    poses = []
    for i in range(B):
        th = 2 * np.pi * i / B
        poses.append(look_at(4.0 * np.array([np.cos(th) * 0.8, np.sin(th) * 0.8, 0.6])))
    pose_gt = torch.tensor(np.stack(poses))
    f = 0.5 * W / np.tan(0.5 * 0.69)
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float32)[None].repeat(B, 1, 1)
    se3 = (torch.randn(B, 6, generator=g) * 0.01).requires_grad_(True)
    noise = O.se3_to_SE3(torch.randn(B, 6, generator=g) * 0.15)
    target = torch.rand(B, r, 3, generator=g)

    def step():
        for _, v in O.flat_params(params):
            v.grad = None
        ray_idx = torch.randperm(H * W, generator=g)[:r]
        pose = O.train_pose(se3, noise, pose_gt)
        c, d = O.rays_for_pixels(pose, intr.inverse(), ray_idx, W)
        jit = torch.rand(B * r, 1, generator=g)
        rgb, _, _ = O.render(cfg, params, c.reshape(-1, 3), d.reshape(-1, 3), S, white_bg=True, jitter=jit)
        loss = O.render_loss(rgb.view(B, r, 3), target) + 8e-5 * O.density_L1(params)
        loss.backward()

    # north_star: "timed on the box's host cores (count stated)".  The path is thousands of small ops and all cores of a big
    # host oversubscribe it, so the sweep goes from 8 threads up to EVERY logical CPU of the host (8, 16, 32, 64, 128, 256, ...,
    # ncpu) and reports the best; the budget is shared, so the larger counts get one or two repetitions each
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64, 128, 256, ncpu) if t <= ncpu} or {ncpu})
    best = None
    t_start = time.time()
    for nt in cands:
        torch.set_num_threads(nt)
        t_w = time.time()
        step()  # warm-up at this thread count
        t_w = time.time() - t_w
        if best is not None and t_w > 3.0 * best[0]:
            continue  # (hopelessly oversubscribed at this count: its warm-up alone took three best-so-far steps)
        times = []
        while len(times) < 9 and (time.time() - t_start) < seconds_budget * (cands.index(nt) + 1) / len(cands):
            t0 = time.time()
            step()
            times.append(time.time() - t0)
        if not times:
            t0 = time.time()
            step()
            times.append(time.time() - t0)
        med = float(np.median(times))
        if best is None or med < best[0]:
            best = (med, nt, times)
    med, nt, times = best
    return dict(value=B * r / med, cores=nt, threads_tried=cands, reps=len(times), median_ms=med * 1e3,
                min_ms=min(times) * 1e3, max_ms=max(times) * 1e3, rays=B * r, host_cpus=ncpu)


def _cpu_c1_protocol(threads, reps=9):
    """BASELINE.md section 3.1 / SURVEY 8(d) for the reference's CPU-sized configuration C1 (grid 64^3, 512 rays drawn over 4
    views, S = 221, VM 16 / 48, MLP 150 -> 64 -> 64 -> 3; fwd + loss + bwd of the parity-pinned torch port): 3-D blur off and on
    (progress 0: sigma = 0.3 x 64 / 3 voxels, 65 taps), each with default denormals and with torch.set_flush_denormal(True),
    one warm-up and then `reps` repetitions INTERLEAVED over the four cases; min / median / max per case."""
    from oracle import tensorf_oracle as O
    g = torch.Generator().manual_seed(0)
    grid, S = [64, 64, 64], 221
    cfg = O.SceneCfg([-1.5] * 3 + [1.5] * 3, grid, [2.0, 6.0])
    params = O.init_params(grid, scale=0.1, bias=0.0, generator=g)
    for _, v in O.flat_params(params):
        v.requires_grad_(True)
    B, r, H, W = 4, 128, 400, 400
    from joint_tensorf_amd.synthetic import look_at
    pose_gt = torch.tensor(np.stack([look_at(4.0 * np.array([np.cos(2 * np.pi * i / B) * 0.8, np.sin(2 * np.pi * i / B) * 0.8, 0.6]))
                                     for i in range(B)]))
    f = 0.5 * W / np.tan(0.5 * 0.69)
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float32)[None].repeat(B, 1, 1)
    se3 = (torch.randn(B, 6, generator=g) * 0.01).requires_grad_(True)
    noise = O.se3_to_SE3(torch.randn(B, 6, generator=g) * 0.15)
    target = torch.rand(B, r, 3, generator=g)
    kern = O.get_kernel(cfg, 0.3, 64)   # c2f_schedule_density[0] = 0.3 at progress 0, density scale 1

    def step(blur):
        for _, v in O.flat_params(params):
            v.grad = None
        ray_idx = torch.randperm(H * W, generator=g)[:r]
        pose = O.train_pose(se3, noise, pose_gt)
        c, d = O.rays_for_pixels(pose, intr.inverse(), ray_idx, W)
        jit = torch.rand(B * r, 1, generator=g)
        rgb, _, _ = O.render(cfg, params, c.reshape(-1, 3), d.reshape(-1, 3), S, white_bg=True, jitter=jit,
                             kernel_density=kern if blur else None, kernel_color=kern if blur else None)
        (O.render_loss(rgb.view(B, r, 3), target) + 8e-5 * O.density_L1(params)).backward()

    torch.set_num_threads(threads)
    cases = [("blur_off", False, False), ("blur_on", True, False), ("blur_off_flush_denormal", False, True),
             ("blur_on_flush_denormal", True, True)]
    times = {name: [] for name, _, _ in cases}
    try:
        for rep in range(reps + 1):
            for name, blur, flush in cases:
                torch.set_flush_denormal(flush)
                t0 = time.time()
                step(blur)
                if rep > 0:   # repetition 0 is the warm-up of every case
                    times[name].append(time.time() - t0)
    finally:
        torch.set_flush_denormal(False)
    out = {}
    for name, ts in times.items():
        out[name] = dict(min_ms=min(ts) * 1e3, median_ms=float(np.median(ts)) * 1e3, max_ms=max(ts) * 1e3,
                         rays_per_s=B * r / float(np.median(ts)))
    return dict(cases=out, rays=B * r, threads=threads, reps=reps, interleaved=True,
                protocol="BASELINE.md 3.1: C1 (64^3, 512 rays over 4 views, S = 221), fwd + loss + bwd, 1 warm-up + %d interleaved "
                         "repetitions per case, torch %s on the host CPUs" % (reps, torch.__version__))


def cpu_baseline(res, S):
    """The reference algorithm on the box's host cores, same run.  `value`: a BOUNDED SAMPLE OF THE BENCH
    WORKLOAD -- the same grid and samples per ray, 4 views x 32 rays per step instead of 100 x ~20 (a step of the full
    ray batch would take minutes on the CPU).  `c1`: the reference's own CPU-sized configuration C1 (64^3, 512 rays,
    S = 221), the figure earlier lines of this file quoted."""
    main = _cpu_oracle_rate([int(v) for v in res], int(S), 32, seconds_budget=24.0)
    c1 = _cpu_oracle_rate([64, 64, 64], 221, 128, seconds_budget=8.0)
    try:
        c1_protocol = _cpu_c1_protocol(c1["cores"])
    except Exception as e:  # keep the line
        c1_protocol = {"error": repr(e)[:300]}
    return dict(value=main["value"], unit="rays/s", cores=main["cores"], kind="port",
                sample="bench workload at reduced ray count: grid %s, S = %d, %d rays per step (4 views x 32), fwd+loss+bwd, "
                       "blur off; best of thread counts %s: %d threads, %d reps, median %.0f ms (min %.0f / max %.0f); host "
                       "has %d logical CPUs" % ("x".join(str(int(v)) for v in res), int(S), main["rays"],
                                                 main["threads_tried"], main["cores"], main["reps"], main["median_ms"],
                                                 main["min_ms"], main["max_ms"], main["host_cpus"]),
                c1=dict(value=c1["value"], unit="rays/s", cores=c1["cores"],
                        sample="C1: grid 64^3, 512 rays x 221 samples, median %.0f ms over %d reps (min %.0f / max %.0f), best of "
                               "thread counts %s" % (c1["median_ms"], c1["reps"], c1["min_ms"], c1["max_ms"], c1["threads_tried"]),
                        protocol=c1_protocol))


def measure_roofline(model, opt, var, reps=20):
    """Average duration of the dominant kernel launch (fused appearance forward: VM gather + basis +
    MLP), measured with HIP events on the launch stream, against its algorithmic gather bytes."""
    from joint_tensorf_amd import ops
    g = model.graph
    tf = g.nerf.tensorf
    with torch.no_grad():
        pose = g.get_pose(opt, var, mode="train")
        # the pixel lattice of a training iteration (model/nerf.py:655-673), offsets at half a stride
        step = g.lattice_step(opt, len(var.idx))
        sx = torch.arange(step // 2, opt.W, step, device=opt.device)
        sy = torch.arange(step // 2, opt.H, step, device=opt.device)
        ray_idx = (sx[None, :] + sy[:, None] * opt.W).reshape(-1)
        center, ray = ops.ray_gen(pose, var.intr_inv, var.intr, ray_idx, opt.W, ndc=bool(opt.camera.ndc))
    probe = ops.KernelProbe(tf, center.reshape(-1, 3), ray.reshape(-1, 3), g.nerf.n_samples,
                            white_bg=bool(opt.nerf.setbg_opaque), ndc=bool(opt.camera.ndc))
    return probe.run(reps)


def instep_roofline(timers, n_comp_app, n_comp_density=16):
    """roofline block of the dominant kernel from the HIP events recorded around its launches INSIDE the timed steps
    (ops.STEP_TIMERS): algorithmic bytes of SURVEY 8(d) (backward: the gather bytes re-read + the same bytes added to the
    gradients) times the shaded samples of every launch, over the summed launch durations."""
    per = 4 * 3 * n_comp_app * 6
    out = {}
    from joint_tensorf_amd._lib import lib as _l
    _split, _mm = _l.jt_shade_bwd_split(), _l.jt_shade_matrix_mode()
    if _split < 0:
        _split = 16 if (n_comp_app < 48 or (_mm & 4)) else 0
    if _split == 0:
        bwd_name = ("k_shade_bwd (fused appearance backward: MLP backward on the fp32 matrix cores + run-length scatter of the "
                    "factor gradients), inside the timed training steps")
    elif _split == 1:
        bwd_name = ("k_shade_bwd<split> + k_tile_bin / scan / scatter / gxyz (appearance backward: MLP backward chain, then the "
                    "TILE-OWNED scatter of the factor gradients), inside the timed training steps")
    else:
        bwd_name = ("k_shade_bwd<split> + k_shade_scatter (appearance backward: MLP backward chain on the %s matrix cores, then the "
                    "run-length scatter of the factor gradients in runs of %d samples; the two launches timed together), inside "
                    "the timed training steps" % ("bf16 (three-piece operands)" if (_mm & 4) else "fp32", _split))
    for kind, mult, name in (("bwd", 2, bwd_name),
                             ("fwd", 1, "k_shade_fwd<train> (gather + basis + MLP + layer-input records), inside the timed "
                                        "training steps"),
                             ("march_bwd", 2, "k_march_bwd_scan + k_march_bwd_walk (density backward: transmittance suffix "
                                              "scan, then the run-length scatter of the density-factor gradients; bytes "
                                              "counted for the walk's listed samples only), inside the timed training steps")):
        if kind == "march_bwd":
            per = 4 * 3 * n_comp_density * 6
        rows = [(a.elapsed_time(b) * 1e-3, int(off[-1])) for k, a, b, off in timers if k == kind]
        if not rows:
            continue
        t, n = sum(r[0] for r in rows), sum(r[1] for r in rows)
        ach = n * mult * per / t / 1e9
        if kind == "bwd":
            # split backward with the weight-gradient GEMMs forked behind the chain: the fork event separates the two launches
            # (single-chunk launches only: with several chunks the event marks the LAST chunk's chain)
            parts = {}
            for sub in ("bwd_chain", "bwd_scatter"):
                rs = [a.elapsed_time(b) * 1e-3 for k, a, b, off in timers if k == sub]
                if rs and len(rs) == len(rows):
                    parts[sub] = sum(rs) / len(rs) * 1e3
        out[kind] = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                     "launch_ms": t / len(rows) * 1e3, "launches": len(rows), "samples_per_launch": n / len(rows),
                     "samples_per_launch_min_max": [min(r[1] for r in rows), max(r[1] for r in rows)],
                     "bytes_per_sample": mult * per, "timing": "HIP events on the launch stream around every launch of the "
                                                                "timed steps"}
        if kind == "bwd" and parts:
            out[kind]["launch_ms_chain"] = parts.get("bwd_chain")
            out[kind]["launch_ms_scatter"] = parts.get("bwd_scatter")
            out[kind]["launch_ms_parts_note"] = ("the fork event behind the chain kernel splits launch_ms: chain = "
                                                 "k_shade_bwd<split>, scatter = k_shade_scatter (the gather + gradient bytes the "
                                                 "roofline counts all move in the scatter; the chain streams records)")
            if n_comp_app >= 48 and os.environ.get("JT_NO_AUX") != "1" and not os.environ.get("JT_SCATTER_WGS"):
                # (jt_shade.hip: scatter_wgs) the scatter shares the chip ON PURPOSE: its launch takes 1.25 ms instead of the
                # 0.96 ms of a 256-workgroup launch, the step 2 % less
                from joint_tensorf_amd._lib import lib as _jl
                lean = bool(_jl.jt_shade_lean_tape()) and _jl.jt_shade_bwd_split() in (-1, 8, 16)
                out[kind]["shares_the_chip"] = (
                    "k_shade_scatter runs on %d of the 256 CUs (%d per XCD) while the %s weight-gradient GEMMs run on the other %d "
                    "from the auxiliary stream: launch_ms is the duration of a launch that has %s of the chip (chain + scatter "
                    "with all of it: extra.default_every_kernel_alone, where the step is slower)"
                    % ((224, 28, "three", 32, "7/8") if (lean and os.environ.get("JT_SCATTER_WAVES") == "8")
                       else (192, 24, "three" if lean else "four", 64, "3/4")))
    return out


ATOMIC_SEGMENTS_PER_S = 20.9e9   # measured: tools/atomic_rate.hip on MI355X (profiles/round2_atomic_rate.txt)


def live_pmc_traffic(steps=6, warmup=2):
    """roofline.traffic measured in THIS run (VERDICT r4 "weak" 9: the figure used to be read from a committed file): two short
    child runs of this very command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace
    only beside the counters, the program itself behind `--`: what the HBM section of MI355X_MICROARCH.md and the pool's rules
    prescribe), summed per kernel by tools/pmc_traffic_instep.py.  Returns its record, or None when the profiler is not there
    or a pass fails (the committed passes are quoted then)."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic_instep as PT
    tmp = tempfile.mkdtemp(prefix="jt_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
             "--no-probe", "--no-torch-baseline", "--no-extras", "--no-live-pmc"]
    env = dict(os.environ, TMPDIR="/tmp", JT_TIME_WALK="1")
    csvs = {}
    line = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, counter)
        r = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                           cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        path = os.path.join(d, "p_counter_collection.csv")
        if r.returncode != 0 or not os.path.exists(path):
            return None
        csvs[counter] = path
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if lines:
            line = lines[-1]
    if line is None:
        return None
    bench_json = os.path.join(tmp, "line.json")
    open(bench_json, "w").write(line + "\n")
    out_json = os.path.join(tmp, "traffic.json")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        PT.main(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"], bench_json, out_json)
    rec = json.load(open(out_json))
    shutil.rmtree(tmp, ignore_errors=True)
    return rec


def pmc_traffic_instep(roof, hidden=None, live=None):
    """HBM-side bytes per launch of k_shade_bwd from rocprofv3 --pmc passes over THIS command (separate FETCH_SIZE / WRITE_SIZE
    runs of `bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras`; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950), per shaded sample, scaled by the samples of the live launches.  `live`: the
    record of live_pmc_traffic() -- collected in this very run; otherwise the committed passes (tools/profile_cmd.sh ->
    tools/pmc_traffic_instep.py) are quoted and the source says so; null when neither exists."""
    tag = "round4" if os.path.exists(os.path.join(ROOT, "profiles", "round4_default_pmc_traffic.json")) else "round3"
    for t in ("round5", "round6"):
        if os.path.exists(os.path.join(ROOT, "profiles", t + "_default_pmc_traffic.json")):
            tag = t
    path = os.path.join(ROOT, "profiles", tag + "_default_pmc_traffic.json")
    try:
        rec = live if live is not None else json.load(open(path))
        k = rec["k_shade_bwd"]
        n = roof["samples_per_launch"]
        if abs(n - rec["process_samples_per_launch"]) > 0.1 * n:
            return {"traffic": None}  # another workload than the one the counters were collected on
        src = ("measured in this run: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) of this command, %d launches "
               "averaged" % rec["k_shade_bwd"]["launches_averaged"]) if live is not None else \
            "profiles/%s_default_pmc_traffic.json" % tag
        out = {"traffic": k["hbm_bytes_per_sample"] * n, "traffic_unit": "bytes/launch",
               "traffic_source": src, "traffic_detail": rec}
        try:
            # matrix-pipe / vector-instruction / texture-path busy fractions of the big kernels and the launches of a step,
            # from the committed SQ / TA counter passes and kernel trace of this same command (tools/pipe_busy.py)
            busy = json.load(open(os.path.join(ROOT, "profiles", tag + "_default_pipe_busy.json")))
            out["pipe_busy"] = dict(busy["kernels"], source="profiles/%s_default_pipe_busy.json" % tag)
            out["launches_per_step_profiled"] = busy["launches_per_step_profiled"]
        except Exception:
            pass
        try:
            # ... and of ONE steady-state iteration, counted in the ordered trace of the same command (tools/profile_cmd.sh):
            # the figure above divides ALL launches of the process (data-set upload, parameter / moment fills, priming) by the steps
            lp = json.load(open(os.path.join(ROOT, "profiles", tag + "_default_launches_per_iteration.json")))
            out["launches_per_iteration"] = dict(lp, source="profiles/%s_default_launches_per_iteration.json" % tag)
        except Exception:
            pass
        if hidden:
            # the second bound of a scatter kernel: the chip retires ~20.9 G float-atomic 64-byte segments per second
            # whatever the access pattern (tools/atomic_rate.hip, profiles/round2_atomic_rate.txt).  Segments of a launch =
            # (WRITE_SIZE bytes - the gradient records and coordinate gradients the kernel stores) / 64.
            # stored (not atomically added) bytes per sample: the chain's record rows GO, G1, GF (+ G2 unless the tape is lean:
            # the dW2 GEMM derives it then) and the coordinate gradients, which the split backward's scatter writes once per
            # plane (store, then two read-add-writes); its dBasis slabs are 3 bytes per sample at this workload and ignored.
            # (DESIGN.md quoted 18.5 M segments per launch for this kernel until round 5: that was round 4's scatter, which
            #  still sent the LINE gradients to global memory; with the line summed in LDS it is the ~15 M this formula reports)
            from joint_tensorf_amd._lib import lib as _jl
            split_on = _jl.jt_shade_bwd_split() != 0
            lean_on = bool(_jl.jt_shade_lean_tape()) and _jl.jt_shade_bwd_split() in (-1, 8, 16)
            stored = 4 * (3 + (1 if lean_on else 2) * hidden + 32) + (36 if split_on else 12)
            seg = (k["write_bytes_per_launch"] / rec["process_samples_per_launch"] - stored) / 64.0 * n
            out["atomic_unit"] = {"segments_per_launch": seg, "rate_segments_per_s": ATOMIC_SEGMENTS_PER_S,
                                  "floor_ms": seg / ATOMIC_SEGMENTS_PER_S * 1e3,
                                  "frac": seg / ATOMIC_SEGMENTS_PER_S * 1e3 / roof["launch_ms"],
                                  "source": "profiles/round2_atomic_rate.txt (rate), WRITE_SIZE pass (segments)"}
        return out
    except Exception:
        return {"traffic": None}


def gpu_clocks():
    """sclk / mclk / power of GPU 0 as rocm-smi reports them right after the timed loop (the driver's box
    and the builder's boxes differ by 4-8 % on the same code; the line should say what the chip was doing).  A dict of the
    fields found, or an error string: never fatal."""
    import shutil
    import subprocess
    # under rocprofv3 the profiler's preloaded library initialises the GPU in every process it is inherited by, and a child that
    # then execs (rocm-smi is a `#!/usr/bin/env python3` script: two hops) is refused on this pool: no clocks in profiled runs
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")) \
            or any(k.startswith("ROCPROF") for k in os.environ):
        return {"skipped": "running under rocprofv3"}
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return {"error": "rocm-smi not found"}
    try:
        r = subprocess.run([smi, "-d", "0", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--showtemp",
                            "--json"], capture_output=True, text=True, timeout=30)
        j = json.loads(r.stdout[r.stdout.index("{"):])
        card = j.get("card0", next(iter(j.values())))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "socclk", "power", "performance level", "junction", "edge")):
                keep[k] = v
        return keep or {"raw": card}
    except Exception as ex:
        return {"error": repr(ex)[:200]}


def torch_gpu_baseline(opt, model, var_all, steps=3):
    """The reference ALGORITHM in stock torch ops on this GPU (the parity-pinned oracle, device=cuda: grid_sample /
    conv1d / cumprod / Linear + autograd + torch.optim.Adam), on the workload just timed: same grid, samples per ray,
    views and lattice.  This is the "reference single-GPU PyTorch rays/s" of BASELINE.json's north_star."""
    from oracle import tensorf_oracle as O
    dev = opt.device
    tf = model.graph.nerf.tensorf
    res, S = model.graph.nerf.resolution, model.graph.nerf.n_samples
    cfg = O.SceneCfg(opt.data.scene_bbox, res, list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio).to(dev)
    sd = {k: v.detach().clone().contiguous() for k, v in tf.state_dict().items()}
    params = O.params_from_state_dict(sd, prefix="")
    leaves = [v for _, v in O.flat_params(params)]
    for v in leaves:
        v.requires_grad_(True)
    B, H, W = len(var_all.idx), opt.H, opt.W
    se3 = torch.zeros(B, 6, device=dev, requires_grad=True)
    noise = model.graph.pose_noise.detach()
    optim = torch.optim.Adam(leaves, lr=1e-2, betas=(0.9, 0.99))
    optim_pose = torch.optim.Adam([se3], lr=1e-3)
    image = var_all.image.view(B, 3, H * W).permute(0, 2, 1)
    step_px = model.graph.lattice_step(opt, B)
    rays = [0]

    def step():
        ox, oy = np.random.randint(step_px), np.random.randint(step_px)
        sx = torch.arange(ox, W, step_px, device=dev)
        sy = torch.arange(oy, H, step_px, device=dev)
        ray_idx = (sx[None, :] + sy[:, None] * W).reshape(-1)
        pose = O.train_pose(se3, noise, var_all.pose)
        c, r = O.rays_for_pixels(pose, var_all.intr_inv, ray_idx, W)
        jit = torch.rand(c.shape[0] * c.shape[1], 1, device=dev)
        rgb, _, _ = O.render(cfg, params, c.reshape(-1, 3), r.reshape(-1, 3), S, white_bg=True, jitter=jit)
        rgb = rgb.view(B, -1, 3)
        loss = O.render_loss(rgb, image[:, ray_idx]) + 8e-5 * O.density_L1(params)
        optim.zero_grad()
        optim_pose.zero_grad()
        loss.backward()
        optim.step()
        optim_pose.step()
        rays[0] += rgb.shape[0] * rgb.shape[1]

    step()
    torch.cuda.synchronize()
    rays[0] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(value=rays[0] / dt, unit="rays/s", ms_per_step=dt / steps * 1e3, steps=steps,
                rays_per_iter=rays[0] / steps, kind="stock torch ops (parity-pinned oracle) on the same GPU, same grid / "
                                                     "samples per ray / views / lattice, fwd+loss+bwd+Adam, blur off")


def run_extras(steps=20, warmup=5):
    """Secondary workloads, each a short child process of this script (its own model, 10 timed steps): numbers DESIGN.md
    quotes, driver-visible here.  The `default_*` children are the HEADLINE workload under another kernel selection: they run
    with the parent's own --steps / --warmup, i.e. on the same host draws -- the same lattice offsets, the same ray and
    shaded-sample counts per iteration -- so that a variant, the default as a child (`default_headline_child`) and the parent's
    own timed loop can be compared to the per cent (the children used to run 10 steps on another lattice)."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-probe",
            "--no-torch-baseline", "--no-extras"]
    same = ["--steps", str(steps), "--warmup", str(warmup)]   # argparse: the later occurrence wins
    cases = [("stage0_blurred", ["--stage", "0"], {}),
             ("stage0_blurred_hipgraph", ["--stage", "0"], {"JT_GRAPH": "1"}),
             ("stage2_blurred_hipgraph", ["--stage", "2"], {"JT_GRAPH": "1"}),
             ("stage4_blurred_it9000", ["--stage", "4", "--it", "9000"], {}),
             ("parent_yaml_299cube_4096rays", ["--n-voxel-final", "27000000", "--n-rays", "4096"], {}),
             ("llff_final_grid", ["--config", "bat_llff_VM_MLP"], {}),
             ("llff_final_grid_it30000_hipgraph", ["--config", "bat_llff_VM_MLP", "--it", "30000"], {"JT_GRAPH": "1"}),
             # the N = 1 point of the strong-scaling curve: the WHOLE 62 500-ray iteration of BASELINE.json configs[3]
             ("configs3_single_gpu", ["--total-rays", "65536", "--steps", "8", "--warmup", "2"], {}),
             # the sharp last stage of a CONVERGED run on the self-consistent scene (few per cent of the samples shaded)
             # (host-bound eager lines: 100 timed steps behind 20 warm-up steps -- the first iterations of a process still build
             #  the optimizer's launch plan and grow the allocator's pools, a visible share of a 20-step window at ~1 ms per step)
             ("fitted_scene_eager", ["--scene", "fitted", "--steps", "100", "--warmup", "20"], {}),
             ("fitted_scene_hipgraph", ["--scene", "fitted"], {"JT_GRAPH": "1"}),
             ("blobs_eager", ["--scene", "blobs", "--steps", "100", "--warmup", "20"], {}),
             ("blobs_hipgraph", ["--scene", "blobs"], {"JT_GRAPH": "1"}),
             # the headline workload with the appearance gradients through the TILE-OWNED scatter of round 5 (chain kernel +
             # binning + one wave per 4 x 4-texel tile on the matrix cores) instead of the fused kernel's run-length atomics:
             # a selectable variant that does not win at this workload; k_shade_bwd_ms = chain + scatter
             # the headline itself, as a child process: parent-vs-child is then separable from variant-vs-variant
             ("default_headline_child", same, {}),
             # round 5's default: the full tape (product rows recorded, dBasis by a fourth GEMM), scatter on 192 CUs
             ("default_round5_full_tape", same, {"JT_LEAN_TAPE": "0"}),
             # the lean tape with the scatter in its eight-wave shape (runs of 16, 224 workgroups): the state before the
             # twelve-wave scatter (three waves per SIMD, runs of 8, 192 workgroups) became the default
             ("default_scatter_8_waves", same, {"JT_SCATTER_WAVES": "8"}),
             ("default_tile_owned_scatter", same, {"JT_BWD_SPLIT": "1"}),
             # ... and through the ONE-kernel backward of rounds 2-4 (fp32 chain + scatter fused, weight gradients on the launch
             # stream): what the default's chain-on-bf16 + scatter + forked weight-gradient GEMMs replaced in round 5
             ("default_fused_backward_kernel", same, {"JT_BWD_SPLIT": "0", "JT_NO_AUX": "1"}),
             # ... and the default's kernels with the WHOLE chip each: the scatter on 256 CUs, the weight-gradient GEMMs behind it
             # on the launch stream, the optimizer as one launch -- k_shade_bwd_ms here is chain + scatter ALONE on the chip
             # (the default's launch shares it with the GEMMs: roofline.shares_the_chip), the step is what sharing buys
             ("default_every_kernel_alone", same, {"JT_SCATTER_WGS": "256", "JT_NO_AUX": "1", "JT_ADAM_EARLY": "0"})]
    out = {}
    for name, flags, env in cases:
        try:
            e = dict(os.environ, JT_TIME_WALK="1")
            e.update(env)
            r = subprocess.run(base + flags, env=e, capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
            out[name] = {"rays_per_s": j["value"], "ms_per_step": j["ms_per_step"], "workload": j["config"]["workload"],
                         "launch": j["config"]["launch"], "steps": j["steps"],
                         "rays_per_iter": j["config"]["rays_per_iter_per_gpu"]}
            for k in ("shaded_samples_per_iter", "nominal_samples_per_iter", "shaded_over_nominal"):
                if k in j["config"]:
                    out[name][k] = j["config"][k]
            if "roofline" in j and "launch_ms" in j["roofline"]:
                out[name]["k_shade_bwd_ms"] = j["roofline"]["launch_ms"]
                out[name]["k_shade_bwd_frac"] = j["roofline"]["frac"]
                db = j["roofline"].get("density_backward")
                if db:
                    out[name]["density_backward_ms"], out[name]["density_backward_frac"] = db["launch_ms"], db["frac"]
                    out[name]["density_backward_listed_samples"] = db["samples_per_launch"]
        except Exception as ex:  # a secondary number must never cost the headline line
            out[name] = {"error": repr(ex)[:200]}
    # BASELINE.json configs[4]: the full 800 x 800 novel-view render at 1 024 depth samples (32 768-ray slices captured as
    # hipGraphs) and a test-time pose-optimisation iteration, dense random-init scene and the sparse blob scene
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "eval_bench.py")
    for name, flags in (("eval_800x800_S1024_hipgraph", ["--graph"]),
                        ("eval_800x800_S1024_blob_scene_hipgraph", ["--graph", "--scene", "blobs"])):
        try:
            r = subprocess.run([sys.executable, tool] + flags, capture_output=True, text=True, timeout=300)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out[name] = {"ms_per_image": j["eval_render"]["ms_per_image"], "rays_per_s": j["eval_render"]["rays_per_s"],
                         "Msamples_per_s": j["eval_render"]["Msamples_per_s"], "launch": j["eval_render"]["launch"],
                         "test_time_pose_optim_ms_per_iter": j["test_time_optim"]["ms_per_iter"],
                         "test_time_pose_optim_rays_per_iter": j["test_time_optim"]["rays_per_iter"]}
        except Exception as ex:
            out[name] = {"error": repr(ex)[:200]}
    # test-time pose optimisation of V held-out views per iteration (Model.evaluate_test_time_photometric_optim_batched): ms per
    # VIEW-iteration, eager, against the serial eager loop of the same process (V = 1) -- the reference runs 200 views x 400
    # iterations one after the other (model/bat.py:265-292)
    for name, flags in (("eval_test_time_optim_batched_dense_scene", []), ("eval_test_time_optim_batched_blob_scene", ["--scene", "blobs"])):
        try:
            r = subprocess.run([sys.executable, tool, "--no-render", "--test-iters", "40", "--batch-views", "1,8,32"] + flags,
                               capture_output=True, text=True, timeout=300)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out[name] = {"ms_per_view_iteration_by_views_per_iteration": j["test_time_optim_batched_ms_per_view_iteration"],
                         "serial_eager_ms_per_iter": j["test_time_optim"]["ms_per_iter"],
                         "rays_per_view_iteration": j["test_time_optim"]["rays_per_iter"]}
        except Exception as ex:
            out[name] = {"error": repr(ex)[:200]}
    for k in ("fitted_scene_eager", "fitted_scene_hipgraph"):
        if k in out and "error" not in out[k]:
            out[k]["note"] = ("the model HOLDS the ground-truth field (4 % of the nominal samples shaded): a lower bound -- the final "
                              "stage of a run TRAINED on that scene measures 2.8 ms per iteration (profiles/round3_full_schedule_"
                              "rendered_scene.jsonl); the dense headline is the representative case")
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not FORCE_DIST:
        sys.exit(launch_ranks(args))   # nothing in this process has touched the GPU
    world, rank, local, backend = setup_dist(args)
    if world != args.gpus and not FORCE_DIST:
        print("bench.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, world), file=sys.stderr)
        sys.exit(3)
    dev = "cuda:%d" % local
    import joint_tensorf_amd  # noqa: F401  (fails loudly without the HIP library)
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views

    torch.manual_seed(0)
    np.random.seed(1234)  # host draws (blur scale) identical on every rank
    opt = make_options(args.config, device=dev)
    if args.n_voxel_final:
        opt.train_schedule.n_voxel_final = args.n_voxel_final
    stage, it0 = stage_setup(opt, args.stage)
    if args.it >= 0:
        it0 = args.it
    if it0 < opt.train_schedule.change_n_rays_after_n_iters:
        opt.nerf.n_rays = opt.train_schedule.n_rays_init
    else:
        opt.nerf.n_rays = opt.train_schedule.n_rays_rest
    if args.n_rays:
        opt.nerf.n_rays = args.n_rays
    # BASELINE.json configs[3]: ONE 65 536-nominal-ray iteration (the GLOBAL count: the lattice stride follows from it,
    # model/nerf.py:660-662), split over the ranks by Graph.ray_shard -- or rendered whole at N = 1 (--total-rays)
    strong = not args.weak and not args.n_rays and (world > 1 or args.total_rays > 0)
    total_rays = (args.total_rays or 65536) if strong else 0
    shard = (args.shard or "pixel") if strong else "offset"
    if strong:
        opt.nerf.n_rays = total_rays
    n_views = int(opt.data.num_views)
    model = build_model(opt, it0, n_views)
    if args.scene == "blobs":
        from joint_tensorf_amd.synthetic import bake_blobs
        bake_blobs(model.graph.nerf.tensorf, n_blobs=12, seed=0)
    var_all = make_views(opt, n_views, seed=0, device=dev)
    if args.scene == "fitted":
        from joint_tensorf_amd.synthetic import gt_scene_for, load_scene_into, render_views
        g_opt, g_graph = gt_scene_for(opt, seed=0)
        with torch.no_grad():
            load_scene_into(model.graph.nerf.tensorf, g_graph.nerf.tensorf)
            var_all.image = render_views(g_opt, g_graph, var_all)   # pictures of THE scene from the ground-truth cameras
            if hasattr(model.graph, "pose_noise"):   # converged: the cameras have been recovered (tests/test_gpu_convergence.py)
                model.graph.pose_noise.copy_(torch.eye(3, 4, device=dev).expand_as(model.graph.pose_noise))
        torch.cuda.empty_cache()
    if world > 1 or FORCE_DIST:
        # the scene gradients are all-reduced inside the renderer's backward, the pose gradients behind it; the host
        # draws (lattice offsets, blur scale) come from NumPy's global generator, seeded identically on every rank
        model.enable_data_parallel(opt, rank, world, shard=shard, force=FORCE_DIST)
        var_all = model.local_views(var_all)       # "view" sharding: this rank's views only
        torch.manual_seed(1000 + rank)              # per-ray jitter: a stream per rank (parameters were built above)
    nerf = model.graph.nerf
    res, S = nerf.resolution, nerf.n_samples
    if args.probe_only:
        from joint_tensorf_amd.options import Opt
        print(json.dumps(measure_roofline(model, opt, Opt(dict(var_all)), reps=10)))
        return

    rays_total = 0
    grad_sums = []
    # JT_GRAPH=1 on one GPU: the steady-state iteration is replayed from a hipGraph (joint_tensorf_amd/graphed.py).
    # Off by default: at this workload the step is GPU-bound (host enqueue 2.1 ms of a 4.8 ms step) and the replay
    # measures the same 4.8 ms.  Data-parallel runs issue their collectives from the eager backward.
    stepper = None
    if world == 1 and not FORCE_DIST and os.environ.get("JT_GRAPH", "0") == "1" \
            and os.environ.get("JT_BENCH_CHECKSUM") != "1":
        from joint_tensorf_amd.graphed import GraphedTrainStep
        stepper = GraphedTrainStep(model, min_repeats=0, adaptive=False)  # the replay itself is what this mode times
    use_graph = [False]
    tape_groups_used = [1]

    from joint_tensorf_amd import ops as jops_b

    def one_step():
        nonlocal rays_total
        from joint_tensorf_amd.options import Opt
        var = Opt(dict(var_all))
        g = model.graph
        if stepper is not None:
            stepper.train_iteration(opt, var, force_eager=not use_graph[0])
            model.after_iteration(opt)
            rgb = stepper.last_var.rgb
            rays_total += rgb.shape[0] * rgb.shape[1]
            return
        g.it = model.it
        model.optim.zero_grad()
        groups = model.tape_groups(opt)
        if groups > 1:
            # an iteration whose tape would not fit the memory budget (configs[3] on ONE GPU: 62 500 rays): forward + backward
            # over ray groups, as Model.train_iteration does it
            tape_groups_used[0] = groups
            loss = model._forward_backward_in_groups(opt, var, groups)
            n_step = int(sum(model.graph._group_rays))
        else:
            var = g.forward(opt, var, mode="train")
            loss = g.compute_loss(opt, var, mode="train")
            loss = model.summarize_loss(opt, var, loss)
            # (render term scaled to its share of the global mean: var.dp_render_scale; the seed is Model.train_iteration's cached ones)
            jops_b.backward(loss.all, gradient=model._backward_seed(loss.all))   # as Model.forward_backward
            n_step = var.rgb.shape[0] * var.rgb.shape[1]
        model.reduce_pose_gradients()
        if os.environ.get("JT_BENCH_CHECKSUM") == "1":
            with torch.no_grad():
                tf = model.graph.nerf.tensorf
                grad_sums.append({
                    "density": float(sum(p.grad.double().abs().sum() for p in tf.density_plane)),
                    "app": float(sum(p.grad.double().abs().sum() for p in tf.app_plane)),
                    "mlp": float(sum(p.grad.double().abs().sum() for p in tf.renderModule.weights())),
                    "basis": float(tf.basis_mat.weight.grad.double().abs().sum()),
                    "se3": float(model.graph.se3_refine.weight.grad.double().abs().sum())})
        if getattr(model.optim, "supports_early_step", False):
            model.optim.step(early_ok=True)   # as Model.end_iteration: nothing touches .grad between backward and step
        else:
            model.optim.step()
        model.optim.zero_grad()
        it = model.it
        model.it += 1
        model.optim_pose.step()
        model.optim_pose.zero_grad()
        if model.sched_pose is not None:
            model.sched_pose.step()
        nerf.set_progress(model.it / opt.max_iter)
        model.after_iteration(opt)
        rays_total += n_step

    def barrier():
        if world > 1 or FORCE_DIST:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def primed_step(offsets):
        seq = list(offsets)  # (ox, oy), handed out cyclically: a step that falls back to the eager path draws again
        orig_randint = np.random.randint

        def fixed(*a, **k):
            seq.append(seq.pop(0))
            return seq[-1]
        np.random.randint = fixed
        try:
            one_step()
        finally:
            np.random.randint = orig_randint

    if not args.no_roofline and stepper is None:
        from joint_tensorf_amd import ops as _jops
        # HIP events around every k_shade_fwd<train> / k_shade_bwd launch; the ones of the timed steps make the roofline
        _jops.STEP_TIMERS = []
        # the density backward is timed as well (one more reduction launch per step) outside the headline workload
        _jops.STEP_TIMERS_WALK = os.environ.get("JT_TIME_WALK") == "1" or args.config != "bat_blender_VM"
    # priming (untimed, in front of the warm-up): one eager step on the densest lattice (offsets 0, 0) so that every
    # persistent workspace and allocator block reaches its final size, then one step per lattice shape so that the
    # hipGraph of each shape is captured before the clock starts
    primed_step([0, 0])
    # JT_BENCH_SAME_STATE=1: the eager run takes the same eight priming iterations, so that an eager and a replayed run time the
    # SAME iterations (same parameters, same host draws) -- the LLFF workload changes with the iteration count and with the draw
    if stepper is not None or (os.environ.get("JT_BENCH_SAME_STATE") == "1" and world == 1 and not FORCE_DIST):
        use_graph[0] = True
        lat = model.graph.lattice_step(opt, n_views)
        for offs in ([0, 0], [0, lat - 1], [lat - 1, 0], [lat - 1, lat - 1]):
            primed_step(offs)   # two consecutive iterations per lattice shape: both parities of the alternating
            primed_step(offs)   # edge-weighted loss (model/tensorf.py:106), each a graph signature of its own
    for w in range(args.warmup):
        one_step()
    barrier()
    rays_total = 0
    from joint_tensorf_amd import ops as jops_t
    # count the C-ABI calls of one step (every call is one to a few kernel launches) -- untimed
    n_calls = [0]
    orig_check = jops_t.check

    def counting_check(rc, what):
        n_calls[0] += 1
        return orig_check(rc, what)
    if stepper is None:  # on every rank: the step contains collectives
        jops_t.check = counting_check
        one_step()
        jops_t.check = orig_check
        barrier()
        rays_total = 0
    n_untimed = len(jops_t.STEP_TIMERS) if jops_t.STEP_TIMERS is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    rays_timed = rays_total
    clocks = gpu_clocks() if rank == 0 else None   # right behind the timed loop: the chip is still in its working state
    all_timers, jops_t.STEP_TIMERS = jops_t.STEP_TIMERS, None
    timers = all_timers[n_untimed:] if all_timers is not None else None
    # JT_BENCH_CHECKSUM=1 (validation of the N > 1 sharding paths against each other): the parameters as the TIMED steps left
    # them -- taken here, in front of the no-collectives leg below, which lets the ranks' parameters drift apart -- and whether
    # every rank holds the same ones (the optimizer step is replicated: identical reduced gradients, identical parameters)
    param_checksum = None
    if os.environ.get("JT_BENCH_CHECKSUM") == "1":
        with torch.no_grad():
            tf = model.graph.nerf.tensorf
            param_checksum = {
                "density": float(sum(p.double().abs().sum() for p in tf.density_plane)),
                "app": float(sum(p.double().abs().sum() for p in tf.app_plane)),
                "mlp": float(sum(p.double().abs().sum() for p in tf.renderModule.weights())),
                "se3": float(model.graph.se3_refine.weight.double().abs().sum())}
            if world > 1 or FORCE_DIST:
                import torch.distributed as dist
                mine = torch.tensor([param_checksum[k] for k in ("density", "app", "mlp", "se3")], device=dev, dtype=torch.float64)
                every = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)
                param_checksum["identical_on_every_rank"] = bool(all(torch.equal(e, every[0]) for e in every))
                param_checksum["ranks_compared"] = len(every)
    # how much of the gradient exchange the backward hid: the same K steps once more with every collective switched off
    # (after the timed region and the checksum; the ranks' parameters drift apart from here on, nothing below reads them)
    dt_nocoll = None
    if world > 1 or FORCE_DIST:
        jops_t._DP["no_collectives"] = True
        one_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        dt_nocoll = time.perf_counter() - t1
        jops_t._DP["no_collectives"] = os.environ.get("JT_DP_NO_COLLECTIVES") == "1"
    # the gradient exchange alone: the three collectives of a backward (appearance factors, density factors, basis +
    # MLP; ops.RenderRays.backward) on buffers of the same sizes, not overlapped with anything, median of 5
    allreduce_ms = None
    if (world > 1 or FORCE_DIST) and jops_t._DP.get("span_elems"):
        import torch.distributed as dist
        allreduce_ms = {}
        for (lo, hi), n in sorted(jops_t._DP["span_elems"].items()):
            buf = torch.zeros(n, device=dev)
            ts = []
            for _ in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                dist.barrier()
                a.record()
                dist.all_reduce(buf, group=jops_t._DP["group"])
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            allreduce_ms["%.1f MB" % (n * 4 / 1e6)] = sorted(ts[1:])[2]
    t = torch.tensor([dt, float(rays_timed), dt_nocoll or 0.0], device=dev, dtype=torch.float64)
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays_all = float(tmax[0]), float(tsum[1])
        dt_nocoll = float(tmax[2])
        per_rank = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, t[1:2].contiguous())
        rays_per_rank = [float(v) / args.steps for v in per_rank]
    else:
        rays_all = float(rays_timed)
        rays_per_rank = [rays_all / args.steps]

    if rank == 0:
        out = {
            "metric": "train rays/sec & Msamples/sec composited, lego VM-48, 1/2/4/8 MI355X",
            "value": rays_all / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            # strong: ONE configs[3] iteration split N ways; weak: the yaml's ray count on every rank; the default N = 1 line is
            # neither (it is the configs[1] headline, NOT the N = 1 point of the strong-scaling curve: see strong_scaling_n1)
            "scaling": "strong" if strong else ("weak" if world > 1 else "n/a"),
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic (%s, random-init appearance factors / MLP, random images, 100 cameras on a radius-4 "
                     "sphere)" % ("random-init density factors" if args.scene == "random"
                                  else "12 opaque Gaussian blobs baked into the density factors")) if args.scene != "fitted" else
                    "synthetic, self-consistent: the model holds the ground-truth field (12 opaque blobs, smooth textured "
                    "appearance) and is supervised with images rendered from it at the ground-truth cameras (100 on a "
                    "radius-4 sphere), cameras at their recovered (ground-truth) poses",
            "config": {
                "workload": "%s stage %d: grid %s, S=%d samples/ray, %d rays/iter/GPU (%s lattice over %d views), "
                            "blur %s, full train step (fwd+loss+bwd+Adam+pose Adam)"
                            % (args.config, stage, "x".join(str(r) for r in res), S, int(rays_all / args.steps / world),
                               opt.nerf.ray_sampling_strategy, n_views,
                               "on" if model.graph.resolve_blur(opt, "vis")[2] else "off"),
                "rays_per_iter_per_gpu": rays_all / args.steps / world,
                "samples_per_ray": S,
                "Msamples_per_s": rays_all * S / dt / 1e6,
                "shade_impl": ({0: "fp32 MFMA", 1: "bf16x3 MFMA forward chain, fp32 MFMA backward chain and weight gradients",
                                2: "fp32 MFMA chains, bf16x3 MFMA weight gradients",
                                3: "bf16x3 MFMA (three-piece operands, fp32-level accuracy) forward chain and weight-gradient "
                                   "GEMMs, fp32 MFMA backward chain"}.get(__import__("joint_tensorf_amd._lib", fromlist=["lib"]).lib.jt_shade_matrix_mode() & 3)
                               + ("; the backward chain of the split appearance backward on bf16x3 MFMA as well"
                                  if __import__("joint_tensorf_amd._lib", fromlist=["lib"]).lib.jt_shade_matrix_mode() & 4 else "")),
                "launch": ("hipGraph replay (%(replayed)d replayed / %(captured)d captured / %(eager)d eager steps)"
                           % stepper.stats) if stepper is not None else "eager",
                "abi_calls_per_step": n_calls[0] or None,
                "ray_groups_per_iteration": tape_groups_used[0],
                "parallelism": ("ray-sharded data parallel x%d (%s shards of ONE %d-nominal-ray iteration = %d rays: the "
                                "ranks together render exactly the single-process iteration), all-reduce of the VM-factor / "
                                "basis / MLP gradients inside the backward + pose gradients behind it"
                                % (world, shard, total_rays, int(round(sum(rays_per_rank))))) if (strong and world > 1) else
                               ("1 GPU, the whole %d-nominal-ray iteration of configs[3] (N = 1 point of the strong-scaling "
                                "curve)" % total_rays if strong else
                                ("ray-sharded data parallel x%d, yaml ray count on every rank, a lattice per rank" % world
                                 if world > 1 else "1 GPU")),
            },
        }
        out["gpu_clocks_after_timed_loop"] = clocks
        if world > 1 or FORCE_DIST:
            import torch.distributed as dist
            out["ranks"] = {"world_size": dist.get_world_size(), "backend": backend,
                            "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0,
                            "devices": "one GPU per rank" if os.environ.get("JT_BENCH_SINGLE_DEVICE") != "1"
                                       else "ALL RANKS ON GPU 0 (functional test, not a scaling measurement)",
                            "shard": shard, "rays_per_iter_per_rank": rays_per_rank,
                            "launcher": "bench.py --gpus N (self-launched ranks)"
                                        if os.environ.get("JT_BENCH_SELF_LAUNCHED") == "1" else "external (torch.distributed.run)"}
        if allreduce_ms is not None:
            out["allreduce_ms"] = allreduce_ms
        if dt_nocoll is not None:
            # step with the collectives minus the same step without them: what of the exchange was NOT hidden behind the
            # backward (the three all-reduces alone, back to back: allreduce_ms)
            out["ms_per_step_no_collectives"] = dt_nocoll / args.steps * 1e3
            out["allreduce_overlap_ms"] = (dt - dt_nocoll) / args.steps * 1e3
        if strong and world > 1:
            # which single-GPU figure this line scales from: the SAME 65 536-nominal-ray iteration on one GPU -- not the
            # default `--gpus 1` headline, whose ~2 000-ray iteration costs 16 % more per ray
            out["strong_scaling_n1"] = {"workload": "python bench.py --gpus 1 --total-rays %d" % total_rays,
                                        "key": "extra.configs3_single_gpu of the default `--gpus 1` line (rays_per_s)",
                                        "note": "scaling efficiency at N = value(N) / (N x that figure); value(N) / value of the "
                                                "default N = 1 line overstates it (different iteration)"}
        if os.environ.get("JT_BENCH_CHECKSUM") == "1":  # validation of the N > 1 paths against each other
            with torch.no_grad():
                tf = model.graph.nerf.tensorf
                out["grad_checksum_first_step"] = grad_sums[0] if grad_sums else None
                out["param_checksum"] = param_checksum
        if not args.no_roofline:
            try:
                from joint_tensorf_amd.options import Opt
                tf_ = model.graph.nerf.tensorf
                ins = instep_roofline(timers or [], tf_.app_n_comp[0], tf_.density_n_comp[0])
                if "bwd" in ins:
                    out["roofline"] = ins["bwd"]
                    # the composited-sample rate counts nominal samples (rays x S); the samples that are
                    # really shaded (in the box and above the weight threshold), read from the launches' device-side counts
                    out["config"]["Msamples_per_s_shaded"] = (ins["bwd"]["samples_per_launch"] * ins["bwd"]["launches"]
                                                              * world / dt / 1e6)
                    out["config"]["shaded_samples_per_iter"] = ins["bwd"]["samples_per_launch"] * ins["bwd"]["launches"] / args.steps
                    out["config"]["nominal_samples_per_iter"] = rays_all / args.steps / world * S
                    out["config"]["shaded_over_nominal"] = (out["config"]["shaded_samples_per_iter"]
                                                            / max(out["config"]["nominal_samples_per_iter"], 1.0))
                    # every k_shade_bwd launch of the process (priming and warm-up included): what a profiler sees
                    alls = [int(off[-1]) for k, a, b, off in all_timers if k == "bwd"]
                    out["roofline"]["process_launches"] = len(alls)
                    out["roofline"]["process_samples_per_launch"] = sum(alls) / max(len(alls), 1)
                    live_rec = None
                    if world == 1 and not FORCE_DIST and not args.no_live_pmc and not args.no_extras \
                            and args.config == "bat_blender_VM" and args.stage < 0 and args.it < 0 and args.scene == "random" \
                            and not args.n_rays and not args.n_voxel_final and not args.total_rays \
                            and os.environ.get("JT_BWD_SPLIT") is None:
                        try:
                            live_rec = live_pmc_traffic()
                        except Exception:
                            live_rec = None
                    out["roofline"].update(pmc_traffic_instep(out["roofline"], int(tf_.renderModule.weights()[2].shape[0]),
                                                              live=live_rec))
                    if "fwd" in ins:
                        out["roofline"]["forward"] = ins["fwd"]
                    if "march_bwd" in ins:
                        out["roofline"]["density_backward"] = ins["march_bwd"]
                else:
                    out["roofline"] = {"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None,
                                       "traffic": None, "note": "no in-step launch timed (hipGraph replay)"}
                if not args.no_probe and not strong:  # the same kernels as isolated back-to-back launches on one fixed lattice batch
                    out["roofline"]["probe"] = measure_roofline(model, opt, Opt(dict(var_all)))
            except Exception as e:  # keep the bench line even if the timing is unavailable
                out["roofline"] = {"error": repr(e)}
        if world == 1 and not strong and not args.no_torch_baseline and args.config == "bat_blender_VM" \
                and args.scene == "random" and not model.graph.resolve_blur(opt, "vis")[2]:
            try:
                from joint_tensorf_amd.options import Opt
                out["torch_gpu_baseline"] = torch_gpu_baseline(opt, model, Opt(dict(var_all)))
                out["vs_baseline"] = out["value"] / out["torch_gpu_baseline"]["value"]
                out["vs_baseline_note"] = ("BASELINE.md holds no published number for this metric; the denominator is the "
                                           "reference algorithm in stock torch ops on the same GPU, measured in this run "
                                           "(torch_gpu_baseline)")
            except Exception as e:
                out["torch_gpu_baseline"] = {"error": repr(e)[:300]}
        if world == 1 and not strong and not args.no_extras:
            # free this process' device memory first: the children build their own models
            del model
            jops_t._WS.clear()
            torch.cuda.empty_cache()
            out["extra"] = run_extras(args.steps, args.warmup)
        # the CPU baseline LAST: its sweep ends on every logical CPU of the host, and the intra-op pool it leaves behind kept
        # the host-bound eager child workloads above 10-15 % slower when it ran in front of them
        if world == 1 and not strong and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res, S)
        print(json.dumps(out))
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist
        dist.barrier()  # rank 0 may still be probing / printing
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
