"""Runs the REAL reference Slice3DDataset (reg_slices/src/datasets.py) on a toy on-disk dataset and stores what
it returns (authoring container only).  The modules the reference imports but this code path never uses
(trimesh, cv2, open3d, h5py) and torchvision.transforms (PIL-backed Resize / ToTensor / Normalize, which is what
torchvision does for PIL inputs) are stubbed.

    python tests/golden/make_golden_dataset.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_import import import_reference_models  # noqa: E402
from slice3d_amd.datasets import write_toy_dataset  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    for name in ("trimesh", "cv2", "open3d", "h5py"):
        sys.modules.setdefault(name, types.ModuleType(name))
    T = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size): self.size = size
        def __call__(self, img): return img.resize(self.size[::-1], Image.BILINEAR)

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)

    class Normalize:
        def __init__(self, mean, std): self.m, self.s = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)
        def __call__(self, t): return (t - self.m) / self.s

    T.Compose, T.Resize, T.ToTensor, T.Normalize = Compose, Resize, ToTensor, Normalize
    sys.modules["torchvision.transforms"] = T
    sys.modules["torchvision"].transforms = T


if __name__ == "__main__":
    import_reference_models()
    install_stubs()
    import importlib
    ds_mod = importlib.import_module("src.datasets")
    rec = {}
    with tempfile.TemporaryDirectory() as tmp:
        write_toy_dataset(tmp, "toy", seed=3)
        for tag, white in (("rgb", False), ("white", True)):
            args = types.SimpleNamespace(n_qry=64, dir_data=tmp, name_dataset="toy", img_size=32, from_which_slices="gt",
                                         use_white_bg=white, n_views=6, categories_train="", categories_test="")
            ds = ds_mod.Slice3DDataset("test", args)
            for i in range(len(ds)):
                for k, v in ds[i].items():
                    rec["%s/%d/%s" % (tag, i, k)] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "dataset_toy_seed3.npz"), **rec)
    print(len(rec), "arrays")
