#!/usr/bin/env python3
"""Listed samples (in-box samples with a non-zero density gradient) per training iteration of the LLFF final stage --
the unit count of k_march_bwd_walk's algorithmic bytes (2 x 1 152 B per listed sample: the density taps re-read + the
same bytes added to the gradients).  Pairs with the rocprofv3 kernel trace of `bench.py --config bat_llff_VM_MLP`
(same seeds, same lattices) for the walk's roofline line in profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import bench
    from joint_tensorf_amd import ops
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "bat_llff_VM_MLP"
    torch.manual_seed(0)
    np.random.seed(1234)
    opt = make_options(cfgname, device="cuda:0")
    stage, it0 = bench.stage_setup(opt, -1)
    opt.nerf.n_rays = opt.train_schedule.n_rays_rest
    n_views = int(opt.data.num_views)
    model = bench.build_model(opt, it0, n_views)
    var = make_views(opt, n_views, seed=0, device="cuda:0")
    rows = []
    for k in range(12):
        loss = model.train_iteration(opt, Opt(dict(var)))
        model.after_iteration(opt)
        torch.cuda.synchronize()
        cfg = model.graph.nerf.tensorf.last_render_cfg
        off, _ = cfg.shade_lists
        R, S = off.numel() - 1, cfg.n_samples
        ws = ops._WS[("cuda:0", "march_bwd")]
        o = (R * S * 4 + 255) // 256 * 256
        o = (o + R * S * 2 + 255) // 256 * 256
        nvalid = ws[o:o + 4 * R].view(torch.int32)
        rows.append(dict(rays=R, samples_per_ray=S, listed=int(nvalid.sum()), shaded=int(off[-1])))
    listed = float(np.mean([r["listed"] for r in rows[2:]]))
    print(json.dumps(dict(config=cfgname, grid=model.graph.nerf.resolution, per_iteration=rows, mean_listed_samples=listed,
                          bytes_per_listed_sample=2 * 4 * 3 * 16 * 6)))


if __name__ == "__main__":
    main()
