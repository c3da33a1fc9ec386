"""Slice3DDataset — the on-disk dataset contract of the reference (reg_slices/src/datasets.py:14-179), host side.

Directory layout (README.md:37-47 of the reference):
    <dir_data>/<name_dataset>/00_img_input/<shape>/<view:03d>.png      RGBA renders + meta.pkl per shape
                              01_img_slices/<shape>/<view:03d>/{X,Y,Z}_{1..4}.png   ground-truth slices (RGBA)
                              02_sdfs/<shape>.npy                      (N,4) float: xyz, sdf
                              03_splits/<category>/<split>.lst
                              04_img_slices_gen / 05_img_slices_rec    generated / reconstructed slices (RGB)
meta.pkl = [K, az[n_views], el[n_views], dist[n_views], cam_poses, scale, offset[3]]
(render_slices/blender_script_input.py:290).

Tensor semantics are the reference's, line for line: RGBA -> RGB*alpha (or white background), bilinear resize
to img_size, [-1,1] normalisation; slice order X1..X4, Z4..Z1, Y1..Y4 (datasets.py:107-121); camera chain
getBlenderProj / get_rotate_matrix(-pi/2) / W2O (datasets.py:124-141, utils.py:29-73,132-171);
sdf = (sdf - 0.003) * scale, points = p * scale + (ox, oz, -oy) (datasets.py:143-151); validation / test
subsample = np.random.seed(1234) permutation (datasets.py:161-165); train view = random, otherwise view 4.

Only numpy, PIL and torch are needed (the reference also imports trimesh / h5py / open3d / cv2 for code paths
this class never reaches).
"""
import os
import pickle
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

_CAM_ROT = np.asarray([[1.910685676922942e-15, 4.371138828673793e-08, 1.0],
                       [1.0, -4.371138828673793e-08, -0.0],
                       [4.371138828673793e-08, 1.0, -4.371138828673793e-08]])


def blender_proj(az, el, distance, img_w=1, img_h=1):
    """utils.py:29-73 getBlenderProj -> (K 3x3, RT 3x4)."""
    f_u = 35.0 * img_w / 32.0
    f_v = 35.0 * img_h / 32.0
    K = np.array(((f_u, 0.0, img_w / 2.0), (0.0, f_v, img_h / 2.0), (0.0, 0.0, 1.0)))
    sa, ca, se, ce = np.sin(-az), np.cos(-az), np.sin(-el), np.cos(-el)
    r_world2obj = np.transpose(np.array(((ca * ce, -sa, ca * se), (sa * ce, ca, sa * se), (-se, 0.0, ce))))
    r_obj2cam = np.transpose(_CAM_ROT)
    r_world2cam = r_obj2cam @ r_world2obj
    t_world2cam = -1.0 * r_obj2cam @ np.array(((distance,), (0.0,), (0.0,)))
    camfix = np.array(((1.0, 0.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, -1.0)))
    return K, np.hstack((camfix @ r_world2cam, camfix @ t_world2cam))


def rotate_matrix(angle):
    """utils.py:132-171 get_rotate_matrix: neg . Rz . Rz . scale_y_neg . Rx (4x4)."""
    c, s = np.cos(angle), np.sin(angle)
    rx = np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]])
    rz = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    sy = np.diag([1.0, -1.0, 1.0, 1.0])
    neg = np.diag([-1.0, -1.0, -1.0, 1.0])
    return np.linalg.multi_dot([neg, rz, rz, sy, rx])


def camera_matrices(az, el, distance):
    """datasets.py:124-141 -> (obj_rot_mat (3,3), trans_mat_wo_rot_tp (4,3)) as float64 arrays."""
    K, RT = blender_proj(az, el, distance, img_w=1, img_h=1)
    rot_full = np.linalg.multi_dot([RT, rotate_matrix(-np.pi / 2)])
    obj_rot_mat = np.transpose(rot_full)[:3, :]
    tmp = np.concatenate((np.eye(3), rot_full[:, 3:4]), axis=1)
    trans = np.linalg.multi_dot([K, tmp, np.eye(4)])
    return obj_rot_mat, np.transpose(trans)


def _to_tensor_normalised(img):
    """T.ToTensor() + T.Normalize(.5,.5): HWC uint8 -> CHW float32 in [-1,1]."""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255.0)
    return (t - 0.5) / 0.5


class Slice3DDataset(Dataset):
    SLICE_ORDER = (("X", "1234"), ("Z", "4321"), ("Y", "1234"))

    def __init__(self, split, args):
        self.split = split
        self.n_qry = args.n_qry
        self.dir_dataset = os.path.join(args.dir_data, args.name_dataset)
        self.name_dataset = args.name_dataset
        self.img_size = args.img_size
        if self.name_dataset == "shapenet":
            cats = args.categories_train if split in ("train", "val") else args.categories_test
            categories = cats.split(",")[:-1]
        else:
            categories = [""]
        self.files = []
        for category in categories:
            with open(os.path.join(self.dir_dataset, "03_splits", category, split + ".lst")) as f:
                self.files += [(category, s) for s in f.read().split()]
        self.dir_sdf = os.path.join(self.dir_dataset, "02_sdfs")
        self.from_which_slices = args.from_which_slices
        self.dir_img_slice = os.path.join(self.dir_dataset, {"gt": "01_img_slices", "gen": "04_img_slices_gen",
                                                             "gt_rec": "05_img_slices_rec"}[self.from_which_slices])
        self.dir_img_ipt = os.path.join(self.dir_dataset, "00_img_input")
        self.use_white_bg = args.use_white_bg
        self.n_views = args.n_views

    def __len__(self):
        return len(self.files)

    # ---- image decoding (datasets.py:73-88, 34) ----
    @staticmethod
    def png_2_whitebg(img):
        a = np.array(img)
        rgb, alpha0 = a[:, :, 0:3], (a[:, :, 3:4] == 0).astype(np.float32)
        return Image.fromarray((np.ones(rgb.shape) * 255 * alpha0 + rgb * (1 - alpha0)).astype(np.uint8))

    @staticmethod
    def png_2_rgb(img):
        a = np.array(img)
        return Image.fromarray((a[:, :, 0:3] * (a[:, :, 3:4] / 255.0)).astype(np.uint8))

    def _rgba(self, img):
        img = self.png_2_whitebg(img) if self.use_white_bg else self.png_2_rgb(img)
        return _to_tensor_normalised(img.resize((self.img_size, self.img_size), Image.BILINEAR))

    def __getitem__(self, index):
        _, shape_id = self.files[index]
        view = random.randint(0, self.n_views - 1) if self.split == "train" else 4
        tag = "%03d" % view
        img_ipt = self._rgba(Image.open(os.path.join(self.dir_img_ipt, shape_id, tag + ".png")))
        slices = []
        for axis, parts in self.SLICE_ORDER:
            for part in parts:
                im = Image.open(os.path.join(self.dir_img_slice, shape_id, tag, "%s_%s.png" % (axis, part)))
                slices.append(_to_tensor_normalised(im) if self.from_which_slices in ("gen", "gt_rec")
                              else self._rgba(im))
        img_slices = torch.cat(slices, 0)

        with open(os.path.join(self.dir_img_ipt, shape_id, "meta.pkl"), "rb") as f:
            meta = pickle.load(f)
        obj_rot_mat, trans_tp = camera_matrices(-meta[1][view], meta[2][view], meta[3][view])
        scale, offset = meta[5], meta[6]

        sdf_npy = np.load(os.path.join(self.dir_sdf, shape_id + ".npy"))
        qry = sdf_npy[:, :3] * scale + np.array([offset[0], offset[2], -offset[1]])
        sdf = (sdf_npy[:, 3] - 0.003) * scale        # the sdfs were extracted at the level of 0.003
        occ = (sdf <= 0).astype(np.float32)
        if self.split == "train":
            np.random.seed()
        else:
            np.random.seed(1234)
        perm = np.random.permutation(len(qry))[:self.n_qry]
        return {
            "img_input": img_ipt,
            "qry_norot": torch.tensor(qry[perm]).float(),
            "obj_rot_mat": torch.tensor(obj_rot_mat).float(),
            "trans_mat_wo_rot_tp": torch.tensor(trans_tp).float(),
            "occ": torch.tensor(occ[perm]).float(),
            "sdf": torch.tensor(sdf[perm]).float(),
            "img_slices": img_slices,
        }


def write_toy_dataset(root, name="toy", shapes=("shape_a", "shape_b"), n_views=6, size=40, n_pts=500, seed=0):
    """Writes a tiny dataset in the reference's on-disk layout (tests, demos)."""
    rng = np.random.default_rng(seed)
    base = os.path.join(root, name)
    for sub in ("00_img_input", "01_img_slices", "02_sdfs", "03_splits"):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
    for split in ("train", "val", "test"):
        with open(os.path.join(base, "03_splits", split + ".lst"), "w") as f:
            f.write("\n".join(shapes) + "\n")
    for sh in shapes:
        os.makedirs(os.path.join(base, "00_img_input", sh), exist_ok=True)
        az = list(rng.uniform(0, 2 * np.pi, n_views))
        el = list(rng.uniform(-0.4, 0.6, n_views))
        dist = list(rng.uniform(1.1, 1.4, n_views))
        meta = [np.eye(3), az, el, dist, np.zeros((n_views, 3, 4)), float(rng.uniform(0.7, 1.1)),
                list(rng.uniform(-0.05, 0.05, 3))]
        with open(os.path.join(base, "00_img_input", sh, "meta.pkl"), "wb") as f:
            pickle.dump(meta, f)
        for v in range(n_views):
            rgba = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
            rgba[: size // 4, :, 3] = 0
            Image.fromarray(rgba, "RGBA").save(os.path.join(base, "00_img_input", sh, "%03d.png" % v))
            d = os.path.join(base, "01_img_slices", sh, "%03d" % v)
            os.makedirs(d, exist_ok=True)
            for axis in "XYZ":
                for part in "1234":
                    rgba = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
                    Image.fromarray(rgba, "RGBA").save(os.path.join(d, "%s_%s.png" % (axis, part)))
        pts = rng.uniform(-0.5, 0.5, (n_pts, 3))
        sdf = np.linalg.norm(pts, axis=1, keepdims=True) - 0.3
        np.save(os.path.join(base, "02_sdfs", sh + ".npy"), np.concatenate([pts, sdf], 1).astype(np.float32))
    return base
