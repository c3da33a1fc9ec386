"""Pre-packed uint8 shards of a Slice3D dataset and their device-side loader (second half).

Why: `Slice3DDataset.__getitem__` (reference reg_slices/src/datasets.py:89-179) decodes 13 RGBA PNGs, composites
alpha, resizes with PIL and loads / permutes the whole SDF point file per sample; at the HIP train step's rate
(~14 samples/s per GPU, 8 GPUs) that is ~1 400 PNG decodes/s on the critical path of 16 worker processes.

`pack_dataset()` runs that image path ONCE, with the reference's own operations (this package's `Slice3DDataset`,
equal to the reference class on the goldens of tests/test_dataset.py), and writes per split

    imgs.npy   uint8   (n_shapes, n_views, 13, S, S, 3)   img_input + the 12 slices (X1..X4, Z4..Z1, Y1..Y4), HWC,
                                                          AFTER alpha compositing and the PIL resize to S
    cams.npy   float32 (n_shapes, n_views, 21)            obj_rot_mat (9) | trans_mat_wo_rot_tp (12)
    pts.npy    float32 (sum N, 4)                         qry_norot xyz | sdf, scaled / offset as datasets.py:143-151
    pts_off.npy int64  (n_shapes + 1)                     row range of every shape in pts.npy
    perm.npy   int32   (sum N)                            np.random.seed(1234) permutation of every shape's points
                                                          (the val / test subset of datasets.py:161-165)
    index.json                                            shape ids, sizes, options the pack was made with

`ShardLoader` memory-maps a split and yields batches that are bit-identical to what the reference class returns
(val / test) or identically distributed (train: random view, random query subset): the uint8 images cross PCIe once
(a quarter of the float bytes) — or not at all when the split is cached in HBM (`cache_on_device=True`; 288 GB per
GPU hold ~100 k samples at 256^2) — and `s3d_dataset_images_fwd` / `s3d_dataset_points_fwd` finish the job on the GPU.
"""
import ctypes as C
import json
import os
import types

import numpy as np
import torch

FILES = ("imgs.npy", "cams.npy", "pts.npy", "pts_off.npy", "perm.npy", "index.json")


def _item_arrays(ds, index, view):
    """One (shape, view) through the reference's image / camera code path, kept as uint8 / float32."""
    import pickle
    from PIL import Image
    from .datasets import camera_matrices
    _, shape_id = ds.files[index]
    tag = "%03d" % view

    def u8(img, rgba):
        if rgba:
            img = ds.png_2_whitebg(img) if ds.use_white_bg else ds.png_2_rgb(img)
            img = img.resize((ds.img_size, ds.img_size), Image.BILINEAR)
        a = np.asarray(img, dtype=np.uint8)
        if a.ndim == 2 or a.shape[2] != 3 or a.shape[0] != ds.img_size or a.shape[1] != ds.img_size:
            raise ValueError("%s: image of shape %s cannot be packed as (%d,%d,3)" % (shape_id, a.shape, ds.img_size,
                                                                                     ds.img_size))
        return a
    imgs = [u8(Image.open(os.path.join(ds.dir_img_ipt, shape_id, tag + ".png")), True)]
    for axis, parts in ds.SLICE_ORDER:
        for part in parts:
            im = Image.open(os.path.join(ds.dir_img_slice, shape_id, tag, "%s_%s.png" % (axis, part)))
            imgs.append(u8(im, ds.from_which_slices not in ("gen", "gt_rec")))
    with open(os.path.join(ds.dir_img_ipt, shape_id, "meta.pkl"), "rb") as f:
        meta = pickle.load(f)
    rot, trans = camera_matrices(-meta[1][view], meta[2][view], meta[3][view])
    cam = np.concatenate([torch.tensor(rot).float().numpy().reshape(-1), torch.tensor(trans).float().numpy().reshape(-1)])
    return np.stack(imgs), cam, meta


def _shape_points(ds, index, meta):
    """All points of a shape as the float32 tensors the reference would build (datasets.py:143-151,167-175)."""
    _, shape_id = ds.files[index]
    scale, offset = meta[5], meta[6]
    sdf_npy = np.load(os.path.join(ds.dir_sdf, shape_id + ".npy"))
    qry = sdf_npy[:, :3] * scale + np.array([offset[0], offset[2], -offset[1]])
    sdf = (sdf_npy[:, 3] - 0.003) * scale
    pts = np.concatenate([torch.tensor(qry).float().numpy(), torch.tensor(sdf).float().numpy()[:, None]], 1)
    np.random.seed(1234)
    perm = np.random.permutation(len(qry)).astype(np.int32)
    return pts.astype(np.float32), perm


def pack_dataset(args, out_dir, splits=("train", "val", "test")):
    """args: the options namespace of reg_slices/options.py (dir_data, name_dataset, img_size, n_views,
    from_which_slices, use_white_bg, categories_*).  Writes out_dir/<split>/{imgs,cams,pts,pts_off,perm}.npy + index.json."""
    from .datasets import Slice3DDataset
    for split in splits:
        a = types.SimpleNamespace(**vars(args))
        a.n_qry = 1
        ds = Slice3DDataset(split, a)
        d = os.path.join(out_dir, split)
        os.makedirs(d, exist_ok=True)
        n, nv, s = len(ds), ds.n_views, ds.img_size
        imgs = np.lib.format.open_memmap(os.path.join(d, "imgs.npy"), mode="w+", dtype=np.uint8, shape=(n, nv, 13, s, s, 3))
        cams = np.zeros((n, nv, 21), dtype=np.float32)
        pts, perms, off = [], [], [0]
        for i in range(n):
            meta = None
            for v in range(nv):
                im, cam, meta = _item_arrays(ds, i, v)
                imgs[i, v] = im
                cams[i, v] = cam
            p, pm = _shape_points(ds, i, meta)
            pts.append(p)
            perms.append(pm)
            off.append(off[-1] + len(p))
        imgs.flush()
        del imgs
        np.save(os.path.join(d, "cams.npy"), cams)
        np.save(os.path.join(d, "pts.npy"), np.concatenate(pts) if pts else np.zeros((0, 4), np.float32))
        np.save(os.path.join(d, "perm.npy"), np.concatenate(perms) if perms else np.zeros((0,), np.int32))
        np.save(os.path.join(d, "pts_off.npy"), np.asarray(off, dtype=np.int64))
        with open(os.path.join(d, "index.json"), "w") as f:
            json.dump({"format": 1, "split": split, "shapes": [s_ for _, s_ in ds.files], "img_size": s, "n_views": nv,
                       "n_slices": 12, "from_which_slices": ds.from_which_slices, "use_white_bg": bool(ds.use_white_bg),
                       "name_dataset": ds.name_dataset}, f)
    return out_dir


class ShardLoader:
    """Batches with the tensor contract of Slice3DDataset + DataLoader(collate), produced on `device`.

        for batch in ShardLoader(dir, "train", batch_size=4, n_qry=100000, device="cuda"): trainer.train_step(batch)

    split 'train': random view per sample (datasets.py:93-94), a random n_qry-subset of the points, shuffled
    shapes (`set_epoch(e)` reseeds; `rank` / `world` shard the shapes like a DistributedSampler, equal length per
    rank); other splits: view 4 and the seed-1234 subset — bit-identical to the reference class.  drop_last as
    train.py:125.  cache_on_device keeps the split's uint8 images and points resident in HBM."""

    def __init__(self, shard_dir, split, batch_size, n_qry, device="cuda", rank=0, world=1, seed=0, drop_last=True,
                 cache_on_device=False, with_occ=True):
        self.dir = os.path.join(shard_dir, split)
        for f in FILES:
            if not os.path.isfile(os.path.join(self.dir, f)):
                raise FileNotFoundError("%s: not a packed split (missing %s; run slice3d_amd.shards.pack_dataset)"
                                        % (self.dir, f))
        self.meta = json.load(open(os.path.join(self.dir, "index.json")))
        self.split, self.bs, self.n_qry = split, int(batch_size), int(n_qry)
        self.device = torch.device(device)
        self.imgs = np.load(os.path.join(self.dir, "imgs.npy"), mmap_mode="r")
        self.cams = torch.from_numpy(np.load(os.path.join(self.dir, "cams.npy")))
        self.pts = np.load(os.path.join(self.dir, "pts.npy"), mmap_mode="r")
        self.perm = np.load(os.path.join(self.dir, "perm.npy"), mmap_mode="r")
        self.off = np.load(os.path.join(self.dir, "pts_off.npy"))
        self.n, self.nv, self.s = self.imgs.shape[0], self.imgs.shape[1], self.imgs.shape[3]
        self.ns = self.imgs.shape[2] - 1
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        self.drop_last, self.with_occ = drop_last, with_occ
        self.files = [("", s_) for s_ in self.meta["shapes"]]
        self._lib = None
        self._dev_imgs = self._dev_pts = self._dev_perm = None
        if cache_on_device:
            self._dev_imgs = torch.from_numpy(np.array(self.imgs)).to(self.device)
            self._dev_pts = torch.from_numpy(np.array(self.pts)).to(self.device)
            self._dev_perm = torch.from_numpy(np.array(self.perm)).to(self.device)
        self._pin = None
        self._pin_done = None   # event after the last H2D copy out of _pin: the next batch must not overwrite it earlier
        self._gen = None        # one device generator, re-seeded per (epoch, rank, batch, item)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _order(self):
        idx = np.arange(self.n)
        if self.split == "train":
            np.random.default_rng((self.seed, self.epoch)).shuffle(idx)
        per = self.n // self.world if self.world > 1 else self.n
        return idx[self.rank::self.world][:per] if self.world > 1 else idx

    def __len__(self):
        n = len(self._order())
        return n // self.bs if self.drop_last else -(-n // self.bs)   # drop_last with n < bs: no batch, as DataLoader

    def __iter__(self):
        order = self._order()
        rng = np.random.default_rng((self.seed, self.epoch, self.rank, 1))
        nb = len(self)
        for b in range(nb):
            ids = order[b * self.bs:(b + 1) * self.bs]
            views = rng.integers(0, self.nv, len(ids)) if self.split == "train" else np.full(len(ids), min(4, self.nv - 1))
            yield self.make_batch(ids, views, batch_no=b)

    def make_batch(self, ids, views, *, batch_no=None):
        """The batch of shapes `ids` seen from `views` (both 1-D integer arrays).  batch_no = its index inside the epoch
        (the iterator passes it): with (seed, epoch, rank, item) it keys the train split's query subsets, so iterating an
        epoch twice — or resuming in its middle — replays the same subsets.  A direct caller (a custom sampler) that has no
        batch index may leave it out: the key is then a hash of `ids` and `views`, so different batches still draw
        different subsets and the same batch replays its own."""
        if batch_no is None:
            import zlib
            batch_no = zlib.crc32(np.asarray(ids, dtype=np.int64).tobytes() + np.asarray(views, dtype=np.int64).tobytes())
        from . import _lib
        if self._lib is None:
            self._lib = _lib.load()
        lib, dev = self._lib, self.device
        b, s, ns = len(ids), self.s, self.ns
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        # images: uint8 across PCIe (or already resident), ToTensor + Normalize on the device
        if self._dev_imgs is not None:
            u8 = torch.stack([self._dev_imgs[int(i), int(v)] for i, v in zip(ids, views)])
        else:
            if self._pin_done is not None:
                self._pin_done.synchronize()      # the previous batch's async upload still reads the staging buffer
            if self._pin is None or self._pin.shape[0] < b:
                self._pin = torch.empty((b, 1 + ns, s, s, 3), dtype=torch.uint8).pin_memory()
            for k, (i, v) in enumerate(zip(ids, views)):
                self._pin[k].copy_(torch.from_numpy(np.array(self.imgs[int(i), int(v)])))
            u8 = self._pin[:b].to(dev, non_blocking=True)
            self._pin_done = torch.cuda.Event()
            self._pin_done.record(torch.cuda.current_stream(dev))
        img_input = torch.empty((b, 3, s, s), dtype=torch.float32, device=dev)
        img_slices = torch.empty((b, 3 * ns, s, s), dtype=torch.float32, device=dev)
        _lib.check(lib.s3d_dataset_images_fwd(u8.data_ptr(), img_input.data_ptr(), img_slices.data_ptr(), b, ns, s, st),
                   "s3d_dataset_images_fwd")
        cam = torch.stack([self.cams[int(i), int(v)] for i, v in zip(ids, views)]).to(dev)
        # query subset
        qry = torch.empty((b, self.n_qry, 3), dtype=torch.float32, device=dev)
        sdf = torch.empty((b, self.n_qry), dtype=torch.float32, device=dev)
        occ = torch.empty((b, self.n_qry), dtype=torch.float32, device=dev) if self.with_occ else None
        for k, i in enumerate(ids):
            lo, hi = int(self.off[int(i)]), int(self.off[int(i) + 1])
            n = hi - lo
            if n < self.n_qry:
                raise ValueError("shape %s has %d points, n_qry = %d" % (self.meta["shapes"][int(i)], n, self.n_qry))
            if self._dev_pts is not None:
                pts = self._dev_pts[lo:hi]
            else:
                pts = torch.from_numpy(np.array(self.pts[lo:hi])).to(dev, non_blocking=True)
            if self.split == "train":      # np.random.seed(); permutation[:n_qry]  ->  a uniform subset, drawn on the device
                # from the loader's own stream: a function of (seed, epoch, rank, batch index, item), so ranks draw
                # different subsets and an epoch iterated twice / resumed replays its own
                if self._gen is None:
                    self._gen = torch.Generator(device=dev)
                self._gen.manual_seed(int(np.random.SeedSequence((self.seed, self.epoch, self.rank, int(batch_no), k))
                                          .generate_state(1, dtype=np.uint64)[0] >> 1))
                idx = torch.randperm(n, device=dev, generator=self._gen)[:self.n_qry].int()
            elif self._dev_perm is not None:
                idx = self._dev_perm[lo:lo + self.n_qry]
            else:
                idx = torch.from_numpy(np.array(self.perm[lo:lo + self.n_qry])).to(dev)
            _lib.check(lib.s3d_dataset_points_fwd(pts.data_ptr(), idx.contiguous().data_ptr(), self.n_qry,
                                                  qry[k].data_ptr(), sdf[k].data_ptr(),
                                                  occ[k].data_ptr() if occ is not None else None, st),
                       "s3d_dataset_points_fwd")
        batch = {"img_input": img_input, "qry_norot": qry, "obj_rot_mat": cam[:, :9].reshape(b, 3, 3).contiguous(),
                 "trans_mat_wo_rot_tp": cam[:, 9:].reshape(b, 4, 3).contiguous(), "sdf": sdf, "img_slices": img_slices}
        if occ is not None:
            batch["occ"] = occ
        return batch


def host_batch(loader, ids, views):
    """CPU restatement of ShardLoader.make_batch for non-train splits (tests without a GPU): numpy only."""
    out = {k: [] for k in ("img_input", "img_slices", "qry_norot", "sdf", "occ", "obj_rot_mat", "trans_mat_wo_rot_tp")}
    for i, v in zip(ids, views):
        u8 = torch.from_numpy(np.array(loader.imgs[int(i), int(v)]))           # (13,S,S,3)
        t = (u8.permute(0, 3, 1, 2).float().div(255.0) - 0.5) / 0.5
        out["img_input"].append(t[0])
        out["img_slices"].append(t[1:].reshape(-1, loader.s, loader.s))
        lo = int(loader.off[int(i)])
        idx = np.asarray(loader.perm[lo:lo + loader.n_qry]).astype(np.int64)
        p = torch.from_numpy(np.array(loader.pts[lo:int(loader.off[int(i) + 1])]))[idx]
        out["qry_norot"].append(p[:, :3].contiguous())
        out["sdf"].append(p[:, 3].contiguous())
        out["occ"].append((p[:, 3] <= 0).float())
        cam = loader.cams[int(i), int(v)]
        out["obj_rot_mat"].append(cam[:9].reshape(3, 3))
        out["trans_mat_wo_rot_tp"].append(cam[9:].reshape(4, 3))
    return {k: torch.stack(v) for k, v in out.items()}
