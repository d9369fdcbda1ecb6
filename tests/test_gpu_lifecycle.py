"""The reference's entry point, step by step (train_3d.py:61-107), against this build's `model/bat_hip.py`:
training sequence (load_dataset -> build_networks -> setup_optimizer -> restore_checkpoint -> setup_visualizer -> train)
and evaluation sequence (load_dataset -> build_networks -> restore_checkpoint -> freeze_scene -> freeze_poses ->
evaluate_full -> generate_videos_synthesis) with the reference's call signatures; and a checkpoint written by the
reference itself restored and rendered."""
import numpy as np
import pytest
import torch

from tests.test_lifecycle import _small_opt, reference_checkpoint_file

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_train_3d_sequence_trains_checkpoints_and_evaluates(tmp_path):
    from joint_tensorf_amd.model import bat_hip as model  # importlib.import_module("model.{}".format(opt.model))
    torch.manual_seed(0)
    np.random.seed(0)
    opt = _small_opt(device=DEV, output_path=str(tmp_path), max_iter=5, camera=dict(noise=0.15),
                     freq=dict(scalar=2, val=100, ckpt=100), optim=dict(test_iter=3))
    # ---- train_3d.py:68-80 ----
    m = model.Model(opt)
    m.load_dataset(opt, train_split="train")
    m.build_networks(opt)
    m.setup_optimizer(opt)
    m.restore_checkpoint(opt)
    m.setup_visualizer(opt)
    grids = []
    orig = m.after_iteration

    def spy(o, it=None):
        orig(o, it)
        grids.append((m.it, m.graph.nerf.tensorf.gridSize.tolist()))
    m.after_iteration = spy
    loss = m.train(opt)
    assert torch.isfinite(loss.all).item()
    # the grid grows right after the training step that makes the counter reach upsample_iters[0] = 2 (SURVEY App. B-19:
    # update_schedule sees the incremented iteration), i.e. after the SECOND step
    assert grids[0] == (1, [10, 10, 10]) and grids[1] == (2, [12, 12, 12]) and grids[-1][0] == 5
    assert m.it == 5 and abs(m.graph.nerf.progress_host - 5 / 5) < 1e-9
    se3_trained = m.graph.se3_refine.weight.detach().clone()
    assert float(se3_trained.abs().sum()) > 0
    # ---- train_3d.py:88-107 ----
    opt2 = _small_opt(device=DEV, output_path=str(tmp_path), max_iter=5, camera=dict(noise=0.15), optim=dict(test_iter=3),
                      load="{0}/model.ckpt".format(str(tmp_path)))
    m2 = model.Model(opt2)
    m2.load_dataset(opt2, eval_split="test", train_split="train")
    m2.build_networks(opt2)
    m2.restore_checkpoint(opt2)
    m2.freeze_scene(opt2)
    m2.freeze_poses(opt2)
    assert m2.graph.nerf.tensorf.gridSize.tolist() == [12, 12, 12]
    assert torch.equal(m2.graph.se3_refine.weight.detach(), se3_trained)
    assert torch.equal(m2.graph.pose_noise, m.graph.pose_noise)
    for (k, a), (_, b) in zip(m.graph.nerf.state_dict().items(), m2.graph.nerf.state_dict().items()):
        assert torch.equal(a, b), k
    res = m2.evaluate_full(opt2)
    assert len(res.psnr_per_view) == len(m2.test_data) == 2 and np.isfinite(res.psnr)
    assert res.R_error.shape == (3,) and res.views[0].rgb_map.shape == (1, 3, 32, 32)
    assert m2.generate_videos_synthesis(opt2) is None


def test_resume_continues_the_run(tmp_path):
    """opt.resume: iteration counter, optimizer moments and lr schedule come back; the resumed run's next losses equal
    the uninterrupted run's (same draws)."""
    from joint_tensorf_amd.model import bat_hip

    def run(n_first, resume):
        torch.manual_seed(0)
        np.random.seed(0)
        opt = _small_opt(device=DEV, output_path=str(tmp_path), max_iter=n_first, camera=dict(noise=0.15),
                         resume=resume, c2f_random_density_blur=False, c2f_alternate_2D_blur=False)
        m = bat_hip.Model(opt)
        m.load_dataset(opt)
        m.build_networks(opt)
        m.setup_optimizer(opt)
        m.restore_checkpoint(opt)
        return opt, m

    opt, m = run(6, False)
    losses = []
    orig = m.train_iteration
    jit = torch.rand(4096, 1, generator=torch.Generator().manual_seed(3)).to(DEV)

    def pinned(o, var):
        np.random.seed(100 + m.it)
        m.graph.nerf.tensorf.jitter_override = jit
        out = orig(o, var)
        losses.append(float(out.all.detach()))
        return out
    m.train_iteration = pinned
    opt.max_iter_run = 6
    m.train(opt)
    full = list(losses)
    # the same run stopped after 3 iterations (early_stop_iter), then resumed
    opt_a, ma = run(6, False)
    opt_a.early_stop_iter = 3
    losses.clear()
    orig_a = ma.train_iteration

    def pinned_a(o, var):
        np.random.seed(100 + ma.it)
        ma.graph.nerf.tensorf.jitter_override = jit
        out = orig_a(o, var)
        losses.append(float(out.all.detach()))
        return out
    ma.train_iteration = pinned_a
    ma.train(opt_a)
    assert len(losses) == 3
    ma.save_checkpoint(opt_a, ep=None, it=3, latest=True)
    opt_b, mb = run(6, True)
    assert mb.iter_start == 3 and mb.graph.nerf.tensorf.gridSize.tolist() == [12, 12, 12]
    orig_b = mb.train_iteration

    def pinned_b(o, var):
        np.random.seed(100 + mb.it)
        mb.graph.nerf.tensorf.jitter_override = jit
        out = orig_b(o, var)
        losses.append(float(out.all.detach()))
        return out
    mb.train_iteration = pinned_b
    mb.train(opt_b)
    assert len(losses) == 6
    np.testing.assert_allclose(losses, full, rtol=2e-5)


def test_a_reference_checkpoint_renders_like_the_reference(tmp_path):
    """tests/golden/reference_checkpoint.npz: written by the reference's util.save_checkpoint after an upsampling,
    restored by the reference into a fresh model and rendered (mode "vis").  This build restores the same file through
    Model.restore_checkpoint and must render the same pixels."""
    path, meta, d = reference_checkpoint_file(tmp_path)
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import Opt
    views = Opt(idx=torch.arange(3), pose=torch.from_numpy(np.array(d["in.pose_gt"])),
                intr=torch.from_numpy(np.array(d["in.intr"])), intr_inv=torch.from_numpy(np.array(d["in.intr_inv"])),
                image=torch.zeros(3, 3, 32, 32))
    opt = _small_opt(device=DEV, load=path, camera=dict(noise=0.15),
                     data=dict(image_size=[32, 32], num_views=3, train_views=views, test_views=views))
    m = bat_hip.Model(opt)
    m.load_dataset(opt, eval_split="test")
    m.build_networks(opt)
    m.restore_checkpoint(opt)
    m.freeze_scene(opt)
    m.freeze_poses(opt)
    g = m.graph
    assert g.nerf.tensorf.gridSize.tolist() == [12, 12, 12] and g.nerf.n_samples == meta["n_samples"]
    g.eval()
    with torch.no_grad():
        var = Opt(dict(m.train_data.all))
        pose = g.get_pose(opt, var, mode="train")
        np.testing.assert_allclose(pose.cpu().numpy(), d["out.current_pose"], atol=2e-6)
        ray_idx = torch.from_numpy(np.array(d["in.ray_idx"])).to(DEV)
        ret = g.render(opt, pose, intr_inv=var.intr_inv, ray_idx=ray_idx, mode="vis", intr=var.intr)
    for k in ("rgb", "opacity"):
        err = np.abs(ret[k].cpu().numpy() - d["out." + k]).max()
        print("reference checkpoint render: max |%s - reference| = %.2e" % (k, err))
        assert err <= 2e-6, (k, err)
    np.testing.assert_allclose(ret["depth"].cpu().numpy(), d["out.depth"], atol=2e-5)


def test_resume_from_a_reference_checkpoint_steps_the_optimizer(tmp_path):
    """opt.resume on the file the REFERENCE wrote: its Adam moments are contiguous NCHW tensors, this
    build's factors are channel-last -- VMAdam.load_state_dict re-lays them by value, and the next optimizer step runs
    and equals torch.optim.Adam's step from the same state."""
    import os
    import shutil
    path, meta, d = reference_checkpoint_file(tmp_path)
    shutil.copy(path, os.path.join(str(tmp_path), "model.ckpt"))
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import Opt
    views = Opt(idx=torch.arange(3), pose=torch.from_numpy(np.array(d["in.pose_gt"])),
                intr=torch.from_numpy(np.array(d["in.intr"])), intr_inv=torch.from_numpy(np.array(d["in.intr_inv"])),
                image=torch.rand(3, 3, 32, 32))
    opt = _small_opt(device=DEV, output_path=str(tmp_path), resume=True, camera=dict(noise=0.15),
                     data=dict(image_size=[32, 32], num_views=3, train_views=views, test_views=views))
    m = bat_hip.Model(opt)
    m.load_dataset(opt)
    m.build_networks(opt)
    m.setup_optimizer(opt)
    m.restore_checkpoint(opt)
    assert m.iter_start == meta["iter"]
    n_state = 0
    ref_params, ref_state = [], {}
    for gi, group in enumerate(m.optim.param_groups):
        for p in group["params"]:
            st = m.optim.state.get(p)
            if not st:
                continue
            n_state += 1
            for k in ("exp_avg", "exp_avg_sq"):
                assert st[k].shape == p.shape and st[k].stride() == p.stride(), (k, st[k].stride(), p.stride())
    assert n_state > 0, "the reference checkpoint carries optimizer state"
    # the saved moments arrived by value
    saved = {int(k.split(".")[2]): None for k in d.files if k.startswith("optim.state.")}
    flat = [p for group in m.optim.param_groups for p in group["params"]]
    for pid in saved:
        np.testing.assert_array_equal(m.optim.state[flat[pid]]["exp_avg"].cpu().numpy(), d["optim.state.%d.exp_avg" % pid])
    # one step from there against torch.optim.Adam on copies
    torch.manual_seed(3)
    clones = [p.detach().clone().contiguous().requires_grad_(True) for p in flat]
    groups = []
    k = 0
    for group in m.optim.param_groups:
        n = len(group["params"])
        groups.append(dict(params=clones[k:k + n], lr=group["lr"]))
        k += n
    ref = torch.optim.Adam(groups, betas=(0.9, 0.99))
    sd = m.optim.state_dict()
    ref.load_state_dict({"state": {i: {kk: (vv.detach().clone().contiguous() if torch.is_tensor(vv) else torch.tensor(float(vv)))
                                        for kk, vv in st.items()} for i, st in sd["state"].items()},
                         "param_groups": ref.state_dict()["param_groups"]})
    for p, c in zip(flat, clones):
        g = torch.randn(p.shape, device=DEV) * 1e-3
        p.grad = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
        c.grad = g.clone()
    m.optim.step()
    ref.step()
    for i, (p, c) in enumerate(zip(flat, clones)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), c.detach().cpu().numpy(), rtol=3e-7, atol=1e-7, err_msg=str(i))


def test_split_iteration_equals_the_whole_iteration(monkeypatch):
    """VERDICT r3 hygiene: an iteration whose tape would exceed the memory budget runs its forward + backward over ray groups
    (Model.tape_groups / _forward_backward_in_groups: pixel shards of the lattice, the photometric mean re-weighted by each
    shard's share, regularisers with the first shard).  Forced here with a tiny budget on a small scene: the gradients and
    the loss of the split iteration equal the unsplit one's."""
    import numpy as np
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views

    def run(groups_wanted):
        torch.manual_seed(0)
        np.random.seed(0)
        opt = make_options("bat_blender_VM", device=DEV, data=dict(image_size=[64, 64], num_views=4),
                           train_schedule=dict(n_voxel_init=24 ** 3, n_rays_init=1024, n_rays_rest=1024, upsample_iters=[10 ** 9]),
                           nerf=dict(n_rays=1024, sample_stratified=False), c2f_mode="None")
        model = bat_hip.Model(opt)
        model.build_networks(opt, n_views=4)
        model.setup_optimizer(opt)
        with torch.no_grad():
            for p in model.graph.nerf.tensorf.density_plane:
                p.mul_(22.0)
        var = make_views(opt, 4, seed=3, device=DEV)
        need = float(opt.nerf.n_rays) * model.graph.nerf.n_samples * 1920.0   # bytes of this iteration's nominal tape
        monkeypatch.setenv("JT_TAPE_BUDGET_GB", repr(16.0 if groups_wanted == 1 else 0.999 * need / groups_wanted / 2 ** 30))
        groups = model.tape_groups(opt)
        np.random.seed(5)
        # gradients of the iteration, before any optimizer step: the body of train_iteration up to the backward
        model.graph.it = model.it
        model.optim.zero_grad()
        model.optim_pose.zero_grad()
        if groups > 1:
            loss = model._forward_backward_in_groups(opt, Opt(dict(var)), groups)
        else:
            v = model.graph.forward(opt, Opt(dict(var)), mode="train")
            loss = model.summarize_loss(opt, v, model.graph.compute_loss(opt, v, mode="train"))
            loss.all.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.graph.named_parameters() if p.grad is not None}
        return groups, float(loss.all), grads, np.random.get_state()[1][:8].tolist()

    g1, l1, a, s1 = run(1)
    g3, l3, b, s3 = run(3)
    assert g1 == 1 and g3 in (3, 4), (g1, g3)
    assert s1 == s3, "the split iteration must leave the host random stream where ONE forward leaves it"
    assert abs(l1 - l3) <= 2e-6 * max(1.0, abs(l1)), (l1, l3)
    assert set(a) == set(b) and len(a) >= 20
    for k in a:
        e = float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-30))
        assert e <= 2e-5, (k, e)
