"""Dataset objects with the interface the reference's Model reads (data/base.py:19-135, data/blender.py:19-91,
data/llff.py:43-97): `len(ds)`, `ds.all` (idx, image [N,3,H,W], pose [N,3,4], intr, intr_inv), `ds.prefetch_all_data`,
`ds.get_all_camera_poses`, `ds.setup_loader` (one view per batch).  No image set can be downloaded here, so the default
is the synthetic scene of SURVEY.md 8(d); `DictDataset` wraps tensors that came from anywhere else (e.g. the reference's
own loaders: pass `ds.all`)."""
import torch

from .options import Opt
from .synthetic import make_views


class DictDataset:
    def __init__(self, opt, all_views, split="train"):
        self.opt, self.split = opt, split
        self.all = Opt({k: v for k, v in dict(all_views).items()})
        n = len(self.all.idx)
        self.list = list(range(n))

    def __len__(self):
        return len(self.list)

    def prefetch_all_data(self, opt):
        return None

    def get_all_camera_poses(self, opt):
        return self.all.pose

    def __getitem__(self, i):
        return {k: (v[i] if torch.is_tensor(v) and v.shape[:1] == (len(self),) else v) for k, v in self.all.items()}

    def setup_loader(self, opt, shuffle=False, drop_last=False):
        """one view per batch, batch dimension kept (torch DataLoader with batch_size 1 in the reference)"""
        out = []
        for i in range(len(self)):
            out.append({k: (v[i:i + 1] if torch.is_tensor(v) and v.shape[:1] == (len(self),) else v)
                        for k, v in self.all.items()})
        return out


class SyntheticDataset(DictDataset):
    """Cameras on a sphere around a random-content scene (Blender) / near-identity forward-facing cameras (LLFF), random
    images: the shapes, value ranges and pose conventions of the real loaders."""

    def __init__(self, opt, split="train", subset=None):
        n = int(opt.data.num_views) if split == "train" else int(opt.data.get("num_test_views", 4))
        if subset:
            n = min(n, int(subset))
        seed = int(opt.get("seed", 0)) + (0 if split == "train" else 1000)
        var = make_views(opt, n, seed=seed, device="cpu")
        super().__init__(opt, var, split)


class RenderedDataset(DictDataset):
    """The self-consistent synthetic scene (`opt.data.synthetic: rendered`): cameras as SyntheticDataset's, images RENDERED
    from one known ground-truth field (synthetic.make_gt_scene) at those cameras by the HIP evaluation renderer -- pictures
    of a scene taken from the poses the run is scored against, which is what the reference's image sets are.  Train and
    test splits look at the same field from different cameras.  Needs the GPU."""

    def __init__(self, opt, split="train", subset=None):
        from .synthetic import make_rendered_views
        n = int(opt.data.num_views) if split == "train" else int(opt.data.get("num_test_views", 4))
        if subset:
            n = min(n, int(subset))
        seed = int(opt.get("seed", 0)) + (0 if split == "train" else 1000)
        var = make_rendered_views(opt, n, seed=seed, device=opt.device)
        super().__init__(opt, var, split)


def load(opt, split, subset=None):
    """Which dataset object a run gets, recorded in `opt.data.dataset_class` and printed:
    * `opt.data.synthetic` = "rendered": RenderedDataset; any other true value ("noise"): SyntheticDataset;
    * otherwise the reference's `data.<dataset>.Dataset` (this build dropped into the reference tree).  Only a MISSING
      loader module falls back to the synthetic noise scene, loudly; a loader that is there and fails (wrong path, missing
      dependency of its own, bad file) raises -- a run must not finish on noise images by accident."""
    syn = opt.data.get("synthetic", False)
    if syn:
        ds = RenderedDataset(opt, split=split, subset=subset) if str(syn) == "rendered" \
            else SyntheticDataset(opt, split=split, subset=subset)
    else:
        import importlib
        name = "data.{}".format(opt.data.dataset)
        try:
            mod = importlib.import_module(name)
        except ImportError as e:
            if getattr(e, "name", None) not in (name, "data"):
                raise          # the loader exists; one of ITS imports is missing
            print("joint_tensorf_amd: WARNING -- no dataset loader `%s` on the path (%s); %s split is the SYNTHETIC NOISE "
                  "scene (set opt.data.synthetic to choose it deliberately)" % (name, e, split))
            mod = None
        ds = mod.Dataset(opt, split=split, subset=subset) if mod is not None \
            else SyntheticDataset(opt, split=split, subset=subset)
    opt.data.dataset_class = type(ds).__module__ + "." + type(ds).__name__
    return ds
