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


def load(opt, split, subset=None):
    """The reference's `data.<dataset>.Dataset` when its package is importable (this build dropped into the reference
    tree) and the image set exists; the synthetic scene otherwise or when opt.data.synthetic is set."""
    if not opt.data.get("synthetic", False):
        try:
            import importlib
            mod = importlib.import_module("data.{}".format(opt.data.dataset))
            return mod.Dataset(opt, split=split, subset=subset)
        except Exception:
            pass
    return SyntheticDataset(opt, split=split, subset=subset)
