"""slice3d_amd.datasets.Slice3DDataset against what the REAL reference dataset class returns on the same toy
on-disk dataset (tests/golden/make_golden_dataset.py; the dataset is regenerated here from the same seed)."""
import os
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN


@pytest.mark.parametrize("tag,white", [("rgb", False), ("white", True)])
def test_dataset_matches_reference(tmp_path, tag, white):
    from slice3d_amd.datasets import Slice3DDataset, write_toy_dataset
    g = np.load(os.path.join(GOLDEN, "dataset_toy_seed3.npz"))
    write_toy_dataset(str(tmp_path), "toy", seed=3)
    args = types.SimpleNamespace(n_qry=64, dir_data=str(tmp_path), name_dataset="toy", img_size=32,
                                 from_which_slices="gt", use_white_bg=white, n_views=6, categories_train="",
                                 categories_test="")
    ds = Slice3DDataset("test", args)
    assert len(ds) == 2
    for i in range(len(ds)):
        item = ds[i]
        assert set(item) == {"img_input", "qry_norot", "obj_rot_mat", "trans_mat_wo_rot_tp", "occ", "sdf", "img_slices"}
        for k, v in item.items():
            want = g["%s/%d/%s" % (tag, i, k)]
            assert tuple(v.shape) == want.shape, k
            assert np.abs(v.numpy() - want).max() <= 1e-6, (k, np.abs(v.numpy() - want).max())
    assert item["img_slices"].shape == (36, 32, 32) and item["trans_mat_wo_rot_tp"].shape == (4, 3)


def test_train_split_randomises_view_and_queries(tmp_path):
    from slice3d_amd.datasets import Slice3DDataset, write_toy_dataset
    write_toy_dataset(str(tmp_path), "toy", seed=1)
    args = types.SimpleNamespace(n_qry=50, dir_data=str(tmp_path), name_dataset="toy", img_size=16,
                                 from_which_slices="gt", use_white_bg=False, n_views=6, categories_train="",
                                 categories_test="")
    ds = Slice3DDataset("train", args)
    a = [ds[0]["qry_norot"] for _ in range(4)]
    assert any(not torch.equal(a[0], x) for x in a[1:])
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True, drop_last=True)
    batch = next(iter(loader))
    assert batch["img_input"].shape == (2, 3, 16, 16) and batch["qry_norot"].shape == (2, 50, 3)


@pytest.mark.parametrize("tag,white", [("rgb", False), ("white", True)])
def test_packed_shards_reproduce_the_reference_dataset(tmp_path, tag, white):
    """slice3d_amd.shards: pack the toy dataset once (PNG decode / compositing / PIL resize at pack time), then the
    packed split's host restatement equals the REAL reference class's tensors (goldens) — bit for bit, for every key."""
    from slice3d_amd.datasets import write_toy_dataset
    from slice3d_amd.shards import ShardLoader, host_batch, pack_dataset
    g = np.load(os.path.join(GOLDEN, "dataset_toy_seed3.npz"))
    write_toy_dataset(str(tmp_path), "toy", seed=3)
    args = types.SimpleNamespace(n_qry=64, dir_data=str(tmp_path), name_dataset="toy", img_size=32,
                                 from_which_slices="gt", use_white_bg=white, n_views=6, categories_train="",
                                 categories_test="")
    out = pack_dataset(args, str(tmp_path / "packed"), splits=("test",))
    ld = ShardLoader(out, "test", batch_size=2, n_qry=64, device="cpu")
    assert len(ld) == 1 and ld.imgs.dtype == np.uint8 and ld.imgs.shape == (2, 6, 13, 32, 32, 3)
    batch = host_batch(ld, [0, 1], [4, 4])
    for i in range(2):
        for k in ("img_input", "qry_norot", "obj_rot_mat", "trans_mat_wo_rot_tp", "occ", "sdf", "img_slices"):
            want = g["%s/%d/%s" % (tag, i, k)]
            assert np.array_equal(batch[k][i].numpy(), want), (k, np.abs(batch[k][i].numpy() - want).max())
