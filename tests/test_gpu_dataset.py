"""Device-side staging of pre-packed shards (s3d_dataset_images_fwd / s3d_dataset_points_fwd through
slice3d_amd.shards.ShardLoader) against the host Slice3DDataset (itself equal to the REAL reference class on the
goldens of tests/test_dataset.py): bit-identical batches for the deterministic splits, valid random subsets for train,
and reg_slices/train.py --shards end to end."""
import glob
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(root, white=False, size=32):
    return types.SimpleNamespace(n_qry=64, dir_data=str(root), name_dataset="toy", img_size=size, from_which_slices="gt",
                                 use_white_bg=white, n_views=6, categories_train="", categories_test="")


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("in_hbm", [False, True])
def test_device_batches_equal_the_host_dataset(tmp_path, white, in_hbm):
    from slice3d_amd.datasets import Slice3DDataset, write_toy_dataset
    from slice3d_amd.shards import ShardLoader, pack_dataset
    write_toy_dataset(str(tmp_path), "toy", shapes=("a", "b", "c"), seed=5, size=48, n_pts=700)
    args = _args(tmp_path, white)
    out = pack_dataset(args, str(tmp_path / "packed"), splits=("val", "train"))
    ds = Slice3DDataset("val", args)
    ld = ShardLoader(out, "val", batch_size=3, n_qry=64, cache_on_device=in_hbm)
    batches = list(ld)
    assert len(batches) == 1
    for i in range(3):
        item = ds[i]
        for k, v in item.items():
            assert torch.equal(batches[0][k][i].cpu(), v), k            # bit-identical to the host class
    # train split: random view (one of the packed views), random subset of the shape's own points, no repeats
    lt = ShardLoader(out, "train", batch_size=2, n_qry=200, cache_on_device=in_hbm, seed=3)
    lt.set_epoch(1)
    seen = []
    for batch in lt:
        assert batch["img_input"].shape == (2, 3, 32, 32) and batch["img_slices"].shape == (2, 36, 32, 32)
        assert batch["qry_norot"].shape == (2, 200, 3) and batch["sdf"].shape == (2, 200)
        assert float(batch["img_input"].abs().max()) <= 1.0
        seen.append(batch["qry_norot"].cpu())
    assert len(seen) == 1                                                # 3 shapes, batch 2, drop_last
    # iterating the same epoch again replays its query subsets (keyed on the batch index inside the epoch, not on a
    # running counter); another epoch draws others
    again = [b["qry_norot"].cpu() for b in lt]
    assert torch.equal(again[0], seen[0])
    lt.set_epoch(2)
    assert not torch.equal(next(iter(lt))["qry_norot"].cpu(), seen[0])
    pts_all = torch.from_numpy(np.load(os.path.join(out, "train", "pts.npy")))
    rows = {tuple(r) for r in pts_all[:, :3].numpy().round(6).tolist()}
    q = seen[0][0].numpy().round(6)
    assert all(tuple(r) in rows for r in q.tolist()) and len({tuple(r) for r in q.tolist()}) == 200


def test_train_py_runs_from_shards(tmp_path):
    from slice3d_amd.datasets import write_toy_dataset
    data = tmp_path / "data"
    write_toy_dataset(str(data), "custom", n_views=6, size=40, n_pts=600, seed=2)
    work = tmp_path / "work"
    work.mkdir()
    common = ["--dir_data", str(data), "--name_dataset", "custom", "--img_size", "32", "--n_qry", "256", "--n_views", "6",
              "--name_exp", "toy", "--shards", str(tmp_path / "packed")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "reg_slices", "pack_shards.py")] + common, cwd=str(work),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "reg_slices", "train.py")] + common +
                       ["--n_bs", "2", "--n_epochs", "2", "--freq_ckpt", "1", "--freq_log", "1", "--mode", "train",
                        "--shards_in_hbm"], cwd=str(work), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "[train]" in r.stdout and "[val]" in r.stdout
    assert len(glob.glob(str(work / "experiments" / "toy" / "ckpt" / "*.ckpt"))) == 2


@pytest.mark.parametrize("tag,white", [("rgb", False), ("white", True)])
@pytest.mark.parametrize("in_hbm", [False, True])
def test_device_batches_equal_the_reference_golden_directly(tmp_path, tag, white, in_hbm):
    """One step instead of two: the batch staged ON THE DEVICE (uint8 shards -> s3d_dataset_images_fwd /
    s3d_dataset_points_fwd) against tests/golden/dataset_toy_seed3.npz, the tensors of the REAL reference
    Slice3DDataset (datasets.py:89-179) on the same toy dataset — bit for bit, every key."""
    from slice3d_amd.datasets import write_toy_dataset
    from slice3d_amd.shards import ShardLoader, pack_dataset
    g = np.load(os.path.join(ROOT, "tests", "golden", "dataset_toy_seed3.npz"))
    write_toy_dataset(str(tmp_path), "toy", seed=3)
    args = _args(tmp_path, white)
    out = pack_dataset(args, str(tmp_path / "packed"), splits=("test",))
    ld = ShardLoader(out, "test", batch_size=2, n_qry=64, cache_on_device=in_hbm, with_occ=True)
    batches = list(ld)
    assert len(batches) == 1
    for i in range(2):
        for k in ("img_input", "qry_norot", "obj_rot_mat", "trans_mat_wo_rot_tp", "occ", "sdf", "img_slices"):
            want = g["%s/%d/%s" % (tag, i, k)]
            got = batches[0][k][i].cpu().numpy()
            assert got.shape == want.shape and np.array_equal(got, want), (k, np.abs(got - want).max())
