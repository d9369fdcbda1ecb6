"""The module-level boundary train_3d.py drives (train_3d.py:61-107): dataset objects, schedule glue, checkpoints in
the reference's file format -- the parts that need no GPU.  (The GPU half is tests/test_gpu_lifecycle.py.)"""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _small_opt(name="bat_blender_VM", device="cpu", **over):
    from joint_tensorf_amd.options import make_options
    kw = dict(data=dict(image_size=[32, 32], num_views=3, num_test_views=2, synthetic=True), camera=dict(noise=False),
              train_schedule=dict(n_voxel_init=10 ** 3, n_voxel_final=16 ** 3, upsample_iters=[2, 50], n_rays_init=60,
                                  n_rays_rest=60), nerf=dict(n_rays=60), max_iter=40000)
    kw.update(over)
    return make_options(name, device=device, **kw)


def test_before_iteration_walks_the_llff_schedule():
    """model/nerf.py:177-205 on bat_llff_VM_MLP: 20 480 rays until iteration 6 000, pose steps every 8 iterations until
    20 000, the pose embedding zeroed at iteration 2 500 -- all applied by Model.before_iteration."""
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import make_options
    opt = make_options("bat_llff_VM_MLP", device="cpu")
    model = bat_hip.Model(opt)
    model.graph = torch.nn.Module()
    model.graph.se3_refine = torch.nn.Embedding(4, 6)
    seen = {}
    for it in (0, 2499, 2500, 2501, 5999, 6000, 19999, 20000, 49999):
        with torch.no_grad():
            model.graph.se3_refine.weight.fill_(1.0)
        model.before_iteration(opt, it)
        seen[it] = (opt.nerf.n_rays, opt.optim.pose_grad_accum_iter, float(model.graph.se3_refine.weight.abs().sum()))
    assert seen[0][:2] == (20480, 8) and seen[5999][:2] == (20480, 8)
    assert seen[6000][:2] == (4096, 8) and seen[19999][:2] == (4096, 8)
    assert seen[20000][:2] == (4096, 1) and seen[49999][:2] == (4096, 1)
    assert seen[2500][2] == 0.0 and all(seen[it][2] == 24.0 for it in seen if it != 2500)
    # the Blender yaml misspells the accumulation key: its schedule stays inert, as in the reference
    optb = make_options("bat_blender_VM", device="cpu")
    mb = bat_hip.Model(optb)
    before = optb.optim.get("pose_grad_accum_iter", None)
    mb.before_iteration(optb, 0)
    assert optb.optim.get("pose_grad_accum_iter", None) == before and optb.nerf.n_rays == 2048


def test_dataset_objects_have_the_reference_interface():
    from joint_tensorf_amd import data as jdata
    opt = _small_opt()
    ds = jdata.load(opt, "train")
    assert len(ds) == 3 and set(ds.all.keys()) >= {"idx", "image", "pose", "intr", "intr_inv"}
    assert ds.get_all_camera_poses(opt).shape == (3, 3, 4) and ds.all.image.shape == (3, 3, 32, 32)
    batches = ds.setup_loader(opt)
    assert len(batches) == 3 and batches[1]["pose"].shape == (1, 3, 4) and int(batches[1]["idx"][0]) == 1
    wrapped = jdata.DictDataset(opt, dict(ds.all))
    assert len(wrapped) == 3


def test_data_load_falls_back_only_when_the_loader_module_is_missing(tmp_path, monkeypatch, capsys):
    """data.load: no `data.<dataset>` module on the path -> the synthetic noise scene, with a warning
    and the class recorded in opt.data.dataset_class; a loader that EXISTS and fails (bad path, missing dependency of
    its own) raises instead of letting the run finish on noise images."""
    import sys
    from joint_tensorf_amd import data as jdata
    for k in [k for k in sys.modules if k == "data" or k.startswith("data.")]:
        monkeypatch.delitem(sys.modules, k)
    opt = _small_opt()
    opt.data.synthetic = False
    ds = jdata.load(opt, "train")
    assert type(ds).__name__ == "SyntheticDataset" and opt.data.dataset_class.endswith("SyntheticDataset")
    assert "SYNTHETIC NOISE" in capsys.readouterr().out
    pkg = tmp_path / "data"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "blender.py").write_text("class Dataset:\n    def __init__(self, opt, split, subset=None):\n"
                                    "        raise FileNotFoundError('no such image set: ' + str(split))\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == "data" or k.startswith("data.")]:
        monkeypatch.delitem(sys.modules, k)
    with pytest.raises(FileNotFoundError):
        jdata.load(opt, "train")
    (pkg / "blender.py").write_text("import a_package_that_is_not_installed\n")
    for k in [k for k in sys.modules if k == "data" or k.startswith("data.")]:
        monkeypatch.delitem(sys.modules, k)
    with pytest.raises(ImportError):
        jdata.load(opt, "train")
    opt.data.synthetic = True      # chosen deliberately: no loader is consulted
    assert type(jdata.load(opt, "train")).__name__ == "SyntheticDataset"


def _build(opt):
    from joint_tensorf_amd.model import bat_hip
    m = bat_hip.Model(opt)
    m.load_dataset(opt, train_split="train")
    m.build_networks(opt)
    m.setup_optimizer(opt)
    return m


def test_checkpoint_roundtrip_after_an_upsampling(tmp_path):
    """save_checkpoint -> a fresh Model -> restore_checkpoint(resume): the scene is resized to the checkpointed grid
    before the state_dict goes in, n_voxel_list / learning rates / Adam moments come back (model/tensorf.py:491-524,
    util.py:120-184)."""
    torch.manual_seed(0)
    opt = _small_opt(output_path=str(tmp_path))
    m = _build(opt)
    nerf = m.graph.nerf
    for p in m.graph.parameters():  # give Adam a state
        if p.requires_grad:
            p.grad = torch.randn_like(p) * 1e-3
    m.optim.step()
    m.optim_pose.step()
    m.it = 2
    nerf.update_schedule(opt, 2)       # upsampling: 10^3 -> 12^3, new optimizer
    nerf.update_schedule(opt, 3)       # one lr decay step
    nerf.set_progress(3 / opt.max_iter)
    grid = nerf.tensorf.gridSize.tolist()
    assert grid == [12, 12, 12] and nerf.n_voxel_list == [4096]
    for p in nerf.tensorf.parameters():
        p.grad = torch.randn_like(p) * 1e-3
    m.optim.step()
    path = m.save_checkpoint(opt, ep=None, it=3, latest=True)
    ck = torch.load(path, weights_only=False)
    assert set(ck) >= {"epoch", "iter", "graph", "opt", "manually_tracked_parameters", "optim", "optim_pose", "sched_pose"}
    assert "nerf.tensorf.density_plane.0" in ck["graph"] and ck["graph"]["nerf.tensorf.app_plane.1"].is_contiguous()
    assert ck["manually_tracked_parameters"]["nerf_reset_kwargs"]["n_voxels"] == nerf.n_voxels

    opt2 = _small_opt(output_path=str(tmp_path), resume=True)
    m2 = _build(opt2)
    assert m2.graph.nerf.tensorf.gridSize.tolist() == [10, 10, 10]
    m2.restore_checkpoint(opt2)
    assert m2.iter_start == 3
    n2 = m2.graph.nerf
    assert n2.tensorf.gridSize.tolist() == grid and n2.n_voxel_list == [4096] and n2.n_samples == nerf.n_samples
    assert abs(n2.lr_index - nerf.lr_index) < 1e-12 and abs(n2.progress_host - 3 / opt.max_iter) < 1e-9
    for (k, a), (_, b) in zip(m.graph.state_dict().items(), m2.graph.state_dict().items()):
        assert torch.equal(a, b), k
    assert [g["lr"] for g in m2.optim.param_groups] == [g["lr"] for g in m.optim.param_groups]
    s1, s2 = m.optim.state_dict()["state"], m2.optim.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) for k in s1)
    # storage stays channel-last after the restore (what the kernels gather from)
    p = n2.tensorf.app_plane[0]
    assert p.permute(0, 2, 3, 1).is_contiguous()


def reference_checkpoint_file(tmp_path):
    """the reference's own checkpoint (tests/golden/reference_checkpoint.npz, written by util.save_checkpoint inside
    tools/make_golden.py) re-packed into the torch file it was"""
    d = np.load(os.path.join(GOLDEN, "reference_checkpoint.npz"))
    meta = json.loads(bytes(d["meta"]).decode())
    graph = {k[len("graph."):]: torch.from_numpy(np.array(d[k])) for k in d.files if k.startswith("graph.")}
    trk = dict(meta["tensorf_reset_kwargs"])
    trk["aabb"] = torch.tensor(trk["aabb"])
    nrk = dict(meta["nerf_reset_kwargs"])
    nrk["bbox"] = torch.tensor(nrk["bbox"])
    state = {}
    for k in d.files:
        if k.startswith("optim.state."):
            _, _, pid, name = k.split(".", 3)
            v = d[k]
            state.setdefault(int(pid), {})[name] = torch.tensor(float(v)) if name == "step" else torch.from_numpy(np.array(v))
    ck = dict(epoch=meta["epoch"], iter=meta["iter"], graph=graph, opt={},
              manually_tracked_parameters={"tensorf_reset_kwargs": trk, "nerf_reset_kwargs": nrk},
              optim=dict(state=state, param_groups=meta["optim_param_groups"]))
    path = os.path.join(str(tmp_path), "reference_model.ckpt")
    torch.save(ck, path)
    return path, meta, d


def test_a_reference_checkpoint_restores_into_this_build(tmp_path):
    path, meta, d = reference_checkpoint_file(tmp_path)
    opt = _small_opt(load=path, camera=dict(noise=0.15), data=dict(image_size=[32, 32], num_views=3, synthetic=True))
    from joint_tensorf_amd.model import bat_hip
    m = bat_hip.Model(opt)
    m.load_dataset(opt)
    # (the pose-noise tensor is built by a HIP kernel: on this CPU-only path the checkpointed one is all that is needed)
    opt.camera.noise = False
    m.build_networks(opt)
    m.graph.pose_noise = torch.nn.Parameter(torch.zeros(3, 3, 4), requires_grad=False)
    m.setup_optimizer(opt)
    m.restore_checkpoint(opt)
    assert m.iter_start == 0 and m.graph.nerf.tensorf.gridSize.tolist() == meta["gridSize"] == [12, 12, 12]
    assert m.graph.nerf.n_samples == meta["n_samples"] and m.graph.nerf.n_voxel_list == meta["nerf_reset_kwargs"]["n_voxel_list"]
    sd = m.graph.state_dict()
    for k in d.files:
        if k.startswith("graph."):
            np.testing.assert_array_equal(sd[k[len("graph."):]].cpu().numpy(), d[k], err_msg=k)
    assert abs(m.optim.param_groups[0]["lr"] - meta["optim_param_groups"][0]["lr"]) < 1e-12
    assert abs(m.graph.nerf.progress_host - meta["progress"]) < 1e-9


def test_train_loop_sequencing_without_a_gpu(tmp_path):
    """nerf.Model.train's control flow (model/nerf.py:150-278) with the device work stubbed out: schedule glue before
    every iteration, update_schedule after it with the incremented counter, validation / checkpoint cadence, early stop,
    and a resumed run that skips the iterations it has already done."""
    from joint_tensorf_amd.options import Opt
    opt = _small_opt(output_path=str(tmp_path), max_iter=12, freq=dict(scalar=4, val=5, ckpt=6), blur_2d=False,
                     train_schedule=dict(n_voxel_init=10 ** 3, n_voxel_final=16 ** 3, upsample_iters=[3, 50], n_rays_init=60,
                                         n_rays_rest=30, change_n_rays_after_n_iters=7))
    m = _build(opt)
    log = []
    m.validate = lambda o, ep=None: log.append(("val", ep))
    m.check_finite = lambda o, loss=None: log.append(("finite", m.it))

    def fake_iteration(o, var):
        log.append(("it", m.it, o.nerf.n_rays, m.graph.nerf.tensorf.gridSize.tolist()[0], tuple(var.image.shape)))
        m.it += 1
        m.graph.nerf.set_progress(m.it / o.max_iter)
        return Opt(all=torch.tensor(0.5))
    m.train_iteration = fake_iteration
    m.train(opt)
    its = [e for e in log if e[0] == "it"]
    assert [e[1] for e in its] == list(range(12))
    assert [e[2] for e in its] == [60] * 7 + [30] * 5                      # n_rays switch at iteration 7
    assert [e[3] for e in its] == [10] * 3 + [12] * 9                      # grid grows after the step that reaches it 3
    assert its[0][4] == (3, 3, 32, 32)
    assert [e[1] for e in log if e[0] == "val"] == [0, 5, 10]             # at the start, then every freq.val
    assert [e[1] for e in log if e[0] == "finite"] == [4, 8, 12, 12]      # every freq.scalar + at the end
    assert sorted(os.listdir(os.path.join(str(tmp_path), "model"))) == ["12.ckpt", "6.ckpt"]
    ck = torch.load(os.path.join(str(tmp_path), "model.ckpt"), weights_only=False)
    assert ck["iter"] == 12
    # resume from the iteration-6 checkpoint: iterations 0..5 are skipped, the grid is the checkpointed one
    opt2 = _small_opt(output_path=str(tmp_path), max_iter=12, freq=dict(scalar=4, val=5, ckpt=100), blur_2d=False, resume=6,
                      train_schedule=dict(n_voxel_init=10 ** 3, n_voxel_final=16 ** 3, upsample_iters=[3, 50], n_rays_init=60,
                                          n_rays_rest=30, change_n_rays_after_n_iters=7))
    m2 = _build(opt2)
    m2.restore_checkpoint(opt2)
    assert m2.iter_start == 6 and m2.graph.nerf.tensorf.gridSize.tolist() == [12, 12, 12]
    log2 = []
    m2.validate = lambda o, ep=None: log2.append(("val", ep))
    m2.check_finite = lambda o, loss=None: None

    def fake2(o, var):
        log2.append(("it", m2.it))
        m2.it += 1
        return Opt(all=torch.tensor(0.5))
    m2.train_iteration = fake2
    opt2.early_stop_iter = 10
    m2.train(opt2)
    assert [e[1] for e in log2 if e[0] == "it"] == [6, 7, 8, 9]
    assert ("val", 0) not in log2  # no start-of-run validation when resuming
