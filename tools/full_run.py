#!/usr/bin/env python3
"""The whole bat_blender_VM training schedule on the synthetic scene, start to finish: every grid stage and
upsampling (optimizer rebuilds), the blur schedule, the 2-D supervision cache refreshed every 500 iterations, edge
masks on alternate iterations until 8000, pose Adam + scheduler.  Reports wall time per schedule segment; asserts a
finite loss throughout.  usage: python tools/full_run.py [--max-iter N] [--config bat_blender_VM]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="bat_blender_VM")
    ap.add_argument("--max-iter", type=int, default=0, help="stop early (0 = the yaml's max_iter)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph", action="store_true", help="replay the sharp-stage iterations from hipGraphs (graphed.py)")
    ap.add_argument("--gpu-time", type=int, default=0, metavar="N",
                    help="every N-th iteration is bracketed by synchronisations and HIP events: GPU time of an "
                         "iteration per segment next to its wall time (host-bound or GPU-bound?)")
    ap.add_argument("--kernel-times", type=int, default=0, metavar="IT",
                    help="HIP-event times of k_shade_fwd / k_shade_bwd / the density backward and their sample counts over "
                         "the 50 iterations from IT (ops.STEP_TIMERS), printed as one JSON line")
    args = ap.parse_args()
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views
    dev = "cuda:0"
    torch.cuda.set_device(0)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    opt = make_options(args.config, device=dev)
    n_views = int(opt.data.num_views)
    model = bat_hip.Model(opt)
    model.build_networks(opt, n_views=n_views)
    model.setup_optimizer(opt)
    views = make_views(opt, n_views, seed=args.seed, device=dev)
    from joint_tensorf_amd import ops
    from joint_tensorf_amd.graphed import GraphedTrainStep
    stepper = GraphedTrainStep(model) if args.graph else None
    last = args.max_iter or int(opt.max_iter)
    marks = sorted(set([0] + [u for u in opt.train_schedule.upsample_iters] + [int(0.3 * opt.max_iter), last]))
    marks = [m for m in marks if m <= last]
    seg, t_seg, it_seg = [], time.perf_counter(), 0
    gpu_ms = []
    worst = 0.0
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for it in range(last):
        model.before_iteration(opt, it)   # ray count, pose-gradient accumulation period, pose resets (model/nerf.py:177-205)
        images, masks, sc = model.select_supervision(opt, views.image)      # model/nerf.py:172-176,209-227
        var = Opt(dict(views))
        var.image, var.train_edge_masks = images, masks
        if args.kernel_times and it == args.kernel_times:
            torch.cuda.synchronize()
            ops.STEP_TIMERS, ops.STEP_TIMERS_WALK = [], True
            t_win = time.perf_counter()
        if args.kernel_times and it == args.kernel_times + 50:
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t_win) / 50 * 1e3
            rows, ops.STEP_TIMERS, ops.STEP_TIMERS_WALK = ops.STEP_TIMERS, None, False
            rep = {"iteration": it, "wall_ms_per_iter": round(wall, 3), "rays": int(opt.nerf.n_rays),
                   "S": model.graph.nerf.n_samples}
            for kind in ("fwd", "bwd", "march_bwd"):
                sel = [(a.elapsed_time(b), int(off[-1])) for k, a, b, off in rows if k == kind]
                if sel:
                    rep[kind] = {"ms": round(sum(x[0] for x in sel) / len(sel), 4),
                                 "samples": round(sum(x[1] for x in sel) / len(sel))}
            print(json.dumps(rep), flush=True)
        timed = args.gpu_time and it % args.gpu_time == args.gpu_time - 1
        if timed:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss = stepper.train_iteration(opt, var) if stepper is not None else model.train_iteration(opt, var)
        if timed:
            e1.record()
            torch.cuda.synchronize()
            gpu_ms.append(e0.elapsed_time(e1))
        model.after_iteration(opt)
        if (it + 1) in marks or (it + 1) % 2000 == 0:
            torch.cuda.synchronize()
            lv = float(loss.all)
            assert np.isfinite(lv), (it, lv)
            model.check_finite(opt)  # the device-side status word: pose / render / loss of every iteration since
            worst = max(worst, lv)
        if (it + 1) in marks:
            now = time.perf_counter()
            tf = model.graph.nerf.tensorf
            seg.append(dict(iters="%d-%d" % (it_seg, it + 1), grid=tf.gridSize.tolist(), S=model.graph.nerf.n_samples,
                            seconds=round(now - t_seg, 2), ms_per_iter=round((now - t_seg) / (it + 1 - it_seg) * 1e3, 3),
                            loss=round(float(loss.all), 5), launch=dict(stepper.stats) if stepper is not None else "eager"))
            if gpu_ms:  # (a synchronised iteration starts on an idle GPU: its span is the GPU work plus the launch ramp)
                seg[-1]["gpu_ms_per_iter_sampled"] = round(float(np.median(gpu_ms)), 3)
                gpu_ms = []
            print(json.dumps(seg[-1]), flush=True)
            t_seg, it_seg = now, it + 1
    torch.cuda.synchronize()
    print(json.dumps(dict(total_seconds=round(time.perf_counter() - t_all, 1), iterations=last,
                          peak_memory_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), max_loss_seen=round(worst, 4))))


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # the failure belongs in the run's record (stdout), not only on stderr
        print(json.dumps({"failed": type(e).__name__, "message": str(e)[:500]}), flush=True)
        raise
