"""Synthetic `feed_dict` generator with the tensor contract of Slice3DDataset.__getitem__
(reference reg_slices/src/datasets.py:89-179): no dataset is reachable offline.

    img_input            (B,3,S,S)        in [-1,1]
    img_slices           (B,3*n_slices,S,S) in [-1,1]
    qry_norot            (B,Q,3)          uniform in [-0.5,0.5]   (datasets.py:171)
    sdf                  (B,Q)            N(0,0.1)
    obj_rot_mat          (B,3,3)          rotation for (az,el) = (-30deg, 20deg)
    trans_mat_wo_rot_tp  (B,4,3)          the dataset's constant camera matrix (f=35/32, d=1.2)

Images are smooth (a few low-frequency sinusoids) rather than white noise: white-noise images make
the feature pyramid vary by O(its own magnitude) between adjacent pixels, which amplifies fp32
rounding of the projected coordinates to ~1e-4 in sdf_pred in ANY fp32 implementation (measured on
the reference itself against an fp64 evaluation); smooth images keep that floor at ~1e-5 so the 1e-4
parity gate is meaningful.  `smooth=False` gives the white-noise variant.
"""
import numpy as np
import torch

TRANS_MAT_WO_ROT_TP = ((1.09375, 0.0, 0.0), (0.0, 1.09375, 0.0), (0.5, 0.5, 1.0), (0.6, 0.6, 1.2))


def smooth_images(rng, n, c, size, nfreq=6):
    yy, xx = np.meshgrid(np.linspace(0, 1, size), np.linspace(0, 1, size), indexing="ij")
    out = np.zeros((n, c, size, size))
    for i in range(n):
        for j in range(c):
            for _ in range(nfreq):
                fx, fy = rng.uniform(-3, 3, 2)
                ph = rng.uniform(0, 2 * np.pi)
                a = rng.uniform(0.1, 0.5)
                out[i, j] += a * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)
    return np.clip(out, -1, 1)


def rotation_az_el(az_deg=-30.0, el_deg=20.0):
    az, el = np.deg2rad(az_deg), np.deg2rad(el_deg)
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    rx = np.array([[1, 0, 0], [0, np.cos(el), -np.sin(el)], [0, np.sin(el), np.cos(el)]])
    return rz @ rx


def make_feed_dict(batch, img_size, n_qry, n_slices=12, seed=0, smooth=True, with_slices=True,
                   device="cpu"):
    rng = np.random.default_rng(seed)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    if smooth:
        img = smooth_images(rng, batch, 3, img_size)
    else:
        img = rng.uniform(-1, 1, (batch, 3, img_size, img_size))
    fd = {"img_input": f32(img)}
    if with_slices:
        if smooth:
            sl = smooth_images(rng, batch, 3 * n_slices, img_size, nfreq=3)
        else:
            sl = rng.uniform(-1, 1, (batch, 3 * n_slices, img_size, img_size))
        fd["img_slices"] = f32(sl)
    fd["qry_norot"] = f32(rng.uniform(-0.5, 0.5, (batch, n_qry, 3)))
    fd["sdf"] = f32(rng.normal(0, 0.1, (batch, n_qry)))
    fd["obj_rot_mat"] = f32(rotation_az_el())[None].repeat(batch, 1, 1).contiguous()
    fd["trans_mat_wo_rot_tp"] = f32(TRANS_MAT_WO_ROT_TP)[None].repeat(batch, 1, 1).contiguous()
    return fd


class SyntheticSlice3DDataset(torch.utils.data.Dataset):
    """Stand-in for Slice3DDataset (datasets.py:14-179) with the same per-sample tensor contract
    (no batch dimension): deterministic in (split, index), sharded over ranks like a DistributedSampler."""

    def __init__(self, length, img_size, n_qry, n_slices=12, split="train", rank=0, world=1):
        per_rank = length // world if world > 1 else length   # equal shard lengths: every rank runs the same number of steps
        self.indices = list(range(rank, length, world))[:max(per_rank, 1)]
        self.img_size, self.n_qry, self.n_slices = img_size, n_qry, n_slices
        self.base = {"train": 0, "val": 1 << 20, "test": 2 << 20}[split]

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        fd = make_feed_dict(1, self.img_size, self.n_qry, self.n_slices, seed=self.base + self.indices[i])
        return {k: v[0] for k, v in fd.items()}


def collate(samples):
    return {k: torch.stack([s[k] for s in samples]) for k in samples[0]}
