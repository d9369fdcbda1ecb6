"""CPU tests of the host side (no GPU needed): option loading, the grid / sample-count schedule of both BAT yamls
against the stage tables SURVEY.md 8(d) reproduced from the reference, the ray lattice, the blur schedule, and that
the product path refuses to run without a GPU instead of falling back."""
import math

import numpy as np
import pytest
import torch

from oracle import tensorf_oracle as O


def _opt(name, **over):
    from joint_tensorf_amd.options import make_options
    return make_options(name, device="cpu", **over)


def test_options_inheritance_and_overrides():
    b = _opt("bat_blender_VM")
    l = _opt("bat_llff_VM_MLP")
    assert b.max_iter == 40000 and l.max_iter == 50000
    assert l.arch.shading.model == "MLP_Fea_WeakView" and b.arch.shading.model == "MLP_Fea"
    assert l.arch.tensorf.density_components == [16, 16, 16]          # inherited through _parent_
    assert l.arch.tensorf.color_components == [20, 20, 20] and b.arch.tensorf.color_components == [48, 48, 48]
    assert l.camera.ndc and not b.camera.ndc and (l.H, l.W) == (480, 640)
    o = _opt("bat_blender_VM", data=dict(image_size=[40, 50]), nerf=dict(n_rays=96))
    assert (o.H, o.W) == (40, 50) and o.nerf.n_rays == 96 and o.nerf.sample_intvs == 1000


@pytest.mark.parametrize("name,table", [
    ("bat_blender_VM", [([64, 64, 64], 221), ([101, 101, 101], 349), ([159, 159, 159], 550), ([252, 252, 252], 872),
                        ([400, 400, 400], 1000)]),
    ("bat_llff_VM_MLP", [([48, 53, 48], 287), ([96, 107, 96], 576), ([192, 214, 192], 1000), ([385, 429, 385], 1000),
                         ([771, 859, 771], 1000)]),
])
def test_grid_schedule_matches_the_reference_stage_table(name, table):
    """resolution and samples per ray of every grid stage, walked through
    NeRF.update_schedule exactly as the training loop does (upsample, optimizer rebuild, sample count)."""
    from joint_tensorf_amd.model import bat_hip
    opt = _opt(name)
    # shrink only the channel counts' memory: the schedule depends on n_voxels and the box, not on the tensors
    nerf = bat_hip.NeRF(opt)
    optim = [nerf._get_optimizer(opt)]
    nerf.get_current_optimizer = lambda: optim[0]
    nerf.register_new_optimizer = lambda o: optim.__setitem__(0, o)
    got = [(nerf.tensorf.gridSize.tolist(), nerf.n_samples)]
    for it in list(opt.train_schedule.upsample_iters):
        if max(table[len(got)][0]) > 260:
            break  # the largest grids are GBs of host memory: checked through the formulas below instead
        nerf.update_schedule(opt, it)
        got.append((nerf.tensorf.gridSize.tolist(), nerf.n_samples))
        assert len(optim[0].param_groups) == 6
    for (g, s), (eg, es) in zip(got, table):
        assert g == eg and s == es, (g, s, eg, es)
    # all stages through the same two formulas the schedule uses (model/tensorf.py:449-461)
    n_list = torch.round(torch.exp(torch.linspace(np.log(opt.train_schedule.n_voxel_init),
                                                  np.log(opt.train_schedule.n_voxel_final), len(table)))).long().tolist()
    scale = [1.0, 1.0, 1.0]
    for i, (eg, es) in enumerate(table):
        sc = opt.train_schedule.resolution_scale_init if i == 0 else scale
        res = O.find_resolution(opt.data.scene_bbox, n_list[i], sc)
        assert res == eg and O.find_n_samples(res, opt.nerf.step_ratio, opt.nerf.sample_intvs) == es


def test_ray_lattice_counts():
    """model/nerf.py:655-673: 2048 rays over 100 Blender views -> step 90 -> 16-25 pixels per view."""
    counts = set()
    for ox in (0, 39, 40, 89):
        for oy in (0, 39, 40, 89):
            idx, step, gh, gw = O.rand_grid_ray_idx(400, 400, 2048, 100, ox, oy)
            assert step == 90 and len(idx) == gh * gw
            counts.add(len(idx) * 100)
    assert min(counts) == 1600 and max(counts) == 2500
    idx, step, _, _ = O.rand_grid_ray_idx(480, 640, 4096, 18, 0, 0)
    assert step == 37 and 3672 <= len(idx) * 18 <= 4212


def test_blur_schedule_switches_off_at_the_yaml_progress():
    from joint_tensorf_amd.model.bat_hip import interp_schedule
    opt = _opt("bat_blender_VM")
    assert interp_schedule(0.0, opt.c2f_schedule_density) == pytest.approx(0.3)
    assert interp_schedule(0.1, opt.c2f_schedule_density) == pytest.approx(0.15)
    assert abs(interp_schedule(0.3, opt.c2f_schedule_density)) < 1e-9 and interp_schedule(0.9, opt.c2f_schedule_density) == 0.0
    assert interp_schedule(0.3, opt.c2f_schedule_density) < 0.001  # under the cut-off of model/tensorf.py:208-220
    assert O.interp_schedule(0.25, opt.c2f_schedule_density) == pytest.approx(interp_schedule(0.25, opt.c2f_schedule_density))


def test_product_path_refuses_to_run_without_a_gpu():
    """No CPU fallback: host tensors are rejected before any kernel argument is built."""
    from joint_tensorf_amd import _lib, ops
    with pytest.raises(_lib.JtError):
        _lib.ptr(torch.zeros(4))
    opt = _opt("bat_blender_VM", data=dict(image_size=[8, 8]), train_schedule=dict(n_voxel_init=8 ** 3))
    from joint_tensorf_amd.model import bat_hip
    nerf = bat_hip.NeRF(opt)
    o = torch.zeros(3, 3)
    d = torch.tensor([[0.0, 0.0, 1.0]]).repeat(3, 1)
    with pytest.raises((AssertionError, RuntimeError, _lib.JtError)):
        nerf.tensorf(opt, o, d, white_bg=True, is_train=False, N_samples=8)
    with pytest.raises(RuntimeError):
        nerf.tensorf.density_L1()


@pytest.mark.parametrize("step,extent", [(90, 400), (64, 400), (16, 400), (17, 640), (37, 480)])
def test_rank_lattice_offsets_keep_the_ray_count(step, extent):
    """Ray-sharded data parallelism: every rank shifts the SHARED lattice draw inside its class, so all ranks
    render the same number of rays in every iteration (no straggler at the gradient all-reduce), rank 0 keeps the
    reference's draw, and the ranks look at different pixels."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("jt_dist", os.path.join(here, "joint_tensorf_amd", "dist.py"))
    jd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(jd)
    for world in (1, 2, 8):
        for o in range(step):
            offs = [jd.rank_lattice_offset(o, step, extent, r, world) for r in range(world)]
            assert offs[0] == o
            assert all(0 <= x < step for x in offs)
            assert len({len(range(x, extent, step)) for x in offs}) == 1
            n_cls = sum(1 for x in range(step) if len(range(x, extent, step)) == len(range(o, extent, step)))
            if n_cls >= world:
                assert len(set(offs)) == world


def test_baked_blob_scene_is_an_exact_vm_field():
    """synthetic.bake_blobs: every Gaussian blob is one rank-1 VM component (plane(x, y) x line(z)) and the last
    component is the constant background -- checked by evaluating the density feature of the baked factors with the
    oracle at the blob centres, between blobs and at a grid corner."""
    import types
    from joint_tensorf_amd.synthetic import bake_blobs
    g = [33, 29, 31]                               # non-cubic on purpose: plane 0 is [1, C, g[1], g[0]], line 0 [1, C, g[2], 1]
    C = 8
    tf = types.SimpleNamespace(
        aabb=torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]]),
        density_plane=[torch.randn(1, C, g[1], g[0]), torch.randn(1, C, g[2], g[0]), torch.randn(1, C, g[2], g[1])],
        density_line=[torch.randn(1, C, g[2], 1), torch.randn(1, C, g[1], 1), torch.randn(1, C, g[0], 1)])
    n = bake_blobs(tf, n_blobs=5, seed=2, amplitude=40.0, radius=(0.3, 0.5), background=-12.0)
    assert n == 5
    cfg = O.SceneCfg(tf.aabb.view(-1).tolist(), g, [2.0, 6.0])
    params = dict(density_plane=tf.density_plane, density_line=tf.density_line)
    # centres the helper drew (same generator, same order of draws)
    rng = np.random.RandomState(2)
    lo, hi = tf.aabb[0], tf.aabb[1]
    centres, sig = [], []
    for _ in range(n):
        centres.append(lo + (hi - lo) * torch.tensor(0.25 + 0.5 * rng.rand(3), dtype=torch.float32))
        sig.append(float(0.3 + 0.2 * rng.rand()))
    pts = torch.stack(centres + [hi.clone(), lo.clone()])
    feat = O.density_feature(cfg, params, O.normalize_coord(cfg, pts))
    expect = []
    for p in pts:
        v = -12.0
        for c, s in zip(centres, sig):
            v += 40.0 * math.exp(-float(((p - c) ** 2).sum()) / (2 * s * s))
        expect.append(v)
    # the grid is coarse (0.1 per texel): bilinear interpolation of a Gaussian of width >= 0.3 is good to a few %
    np.testing.assert_allclose(feat.numpy(), np.array(expect), rtol=0.08, atol=0.3)
    assert float(feat[-1]) < -11.0 and float(feat[:n].min()) > 20.0


def test_bench_gpus_n_without_n_gpus_refuses():
    """`python bench.py --gpus 8` on a machine with fewer GPUs: no line for another N, a non-zero exit (round 2's script
    ignored --gpus and benchmarked one GPU)."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("8 GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "JT_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 8" in r.stderr and not r.stdout.strip()


def test_flat_gradient_buffer_zeroes_only_what_nobody_overwrites():
    """ops._zeros_flat carves the gradient tensors of a backward out of one buffer; groups the regularisers' backward is
    about to overwrite completely (RenderRays.backward: density factors, appearance planes under TV) are left unfilled, every
    other group -- and the padding between its tensors -- is zero, and the spans are what the data-parallel reducer slices."""
    from joint_tensorf_amd import ops
    g = torch.Generator().manual_seed(1)
    groups = [[torch.randn(5, 4, generator=g), torch.randn(3, generator=g)], [torch.randn(8, generator=g)],
              [torch.randn(2, 3, generator=g)], [torch.randn(7, generator=g), torch.randn(1, generator=g)]]
    views, flat, spans = ops._zeros_flat(groups, with_flat=True)
    assert float(flat.abs().sum()) == 0.0 and spans == [(0, 24), (24, 32), (32, 40), (40, 52)]
    for grp, vs in zip(groups, views):
        assert [tuple(v.shape) for v in vs] == [tuple(t.shape) for t in grp]
    # views alias the flat buffer, 16-byte aligned
    views[3][1].fill_(2.0)
    assert float(flat[48]) == 2.0 and all(v.data_ptr() % 16 == 0 for vs in views for v in vs)
    # poison the allocator's next block, then ask for groups 0 and 2 unzeroed
    junk = torch.full((52,), float("nan"))
    del junk
    views, flat, spans = ops._zeros_flat(groups, with_flat=True, unzeroed=(0, 2))
    assert float(flat[24:32].abs().sum()) == 0.0 and float(flat[40:52].abs().sum()) == 0.0
    assert spans == [(0, 24), (24, 32), (32, 40), (40, 52)]
    # the caller overwrites every ELEMENT of the unzeroed groups (their padding words, if any, belong to nobody: the factor
    # tensors this is used for have none, and the data-parallel path, which ships whole spans, never skips a fill)
    for vs in (views[0], views[2]):
        for v in vs:
            v.fill_(1.0)
    assert float(flat[24:32].abs().sum()) == 0.0 and float(flat[40:52].abs().sum()) == 0.0
    assert float(sum(v.sum() for vs in (views[0], views[2]) for v in vs)) == 5 * 4 + 3 + 2 * 3
