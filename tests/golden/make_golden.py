"""Generates tests/golden/*.npz by running the REAL reference (imported from /root/reference through
oracle/ref_import.py) in the authoring container.  Run from the repo root:

    python tests/golden/make_golden.py

The fixtures are data only: inputs + the reference's outputs (and intermediate tensors captured with
forward hooks on the reference modules).  Weights are NOT stored: they are regenerated from
slice3d_amd.weights.seeded_array(key, shape, seed=0) on both sides.  Nothing here travels to the GPU
box except the .npz files.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.ref_import import REG_SLICES, build_reference_model  # noqa: E402
from slice3d_amd.synth import make_feed_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_PROBE = 64  # queries for which stage tensors are stored


def run_case(name, batch, size, n_qry, n_slices, mode, seed, with_stages=False, store_slices=True):
    torch.manual_seed(0)
    model = build_reference_model(n_slices=n_slices, mode=mode, seed=0)
    fd = make_feed_dict(batch, size, n_qry, n_slices, seed=seed)
    captured = {}
    hooks = []
    if with_stages:
        def grab(key):
            def fn(mod, inp, out):
                captured[key] = out.detach().clone()
            return fn
        hooks.append(model.fc_s.register_forward_hook(grab("fc_s")))
        hooks.append(model.fc_p.register_forward_hook(grab("fc_p")))
        for i, layer in enumerate(model.att_decoder.layers):
            hooks.append(layer.register_forward_hook(grab("layer%d" % i)))
    with torch.no_grad():
        feats, _ = model.slices_generator(fd["img_input"])
        out = model({k: v.clone() for k, v in fd.items()})
        qry_rot = fd["qry_norot"].clone()
        if mode == "test":
            qry_rot[:, :, 1:] *= -1
        else:
            qry_rot = torch.bmm(qry_rot, fd["obj_rot_mat"])
        img_pts = model.project_coord(qry_rot, fd["trans_mat_wo_rot_tp"])
    for h in hooks:
        h.remove()
    rec = {
        "meta": np.array([batch, size, n_qry, n_slices, seed], dtype=np.int64),
        "mode": np.array(mode),
        "img_input": fd["img_input"].numpy(),
        "qry_norot": fd["qry_norot"].numpy(),
        "obj_rot_mat": fd["obj_rot_mat"].numpy(),
        "trans_mat_wo_rot_tp": fd["trans_mat_wo_rot_tp"].numpy(),
        "sdf_pred": out["sdf_pred"].numpy(),
        "slices_rec_strided": out["slices_rec"][:, :, ::4, ::4].contiguous().numpy(),
        "vgg_loss": out["vgg_loss"].numpy(),
        "img_pts": img_pts.numpy(),
    }
    if store_slices:
        rec["img_slices"] = fd["img_slices"].numpy()
    # pyramid probes: 256 fixed flat indices per level (NCHW flattening of the reference tensors)
    rng = np.random.default_rng(99)
    for l, f in enumerate(feats):
        idx = rng.integers(0, f.numel(), 256)
        rec["pyr%d_idx" % l] = idx
        rec["pyr%d_val" % l] = f.reshape(-1)[torch.from_numpy(idx)].numpy()
        rec["pyr%d_shape" % l] = np.array(f.shape, dtype=np.int64)
    if with_stages:
        # one sampled level (level 2) for the first N_PROBE queries of every slice row
        pts = img_pts.view(batch, 1, n_qry, 2).expand(-1, n_slices, -1, -1).reshape(batch * n_slices, n_qry, 2)
        s2 = model.sample_from_planes(feats[2], pts[:, :16]).squeeze(1)
        rec["sample_l2"] = s2.numpy()                                   # (B*ns, 16, 128)
        rec["fc_s_rows"] = captured["fc_s"][:N_PROBE].numpy()           # (N_PROBE, ns, 128)
        rec["fc_p_rows"] = captured["fc_p"].reshape(-1, 128)[:N_PROBE].numpy()
        for i in range(3):
            rec["layer%d_tok0" % i] = captured["layer%d" % i][:N_PROBE, 0, :].numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("%-34s sdf|max| %.4f  vgg %.6f  %.1f KB" % (name, float(out["sdf_pred"].abs().max()),
                                                      float(out["vgg_loss"]), os.path.getsize(path) / 1024))


def train_case(name, batch, size, n_qry, n_slices, seed):
    """G4: one train-mode forward/backward of the REAL reference (dropout forced to 0 so the run is
    deterministic; BatchNorm uses batch statistics) with the losses of train.py:41-47."""
    import torch.nn as nn
    import torch.nn.functional as F
    model = build_reference_model(n_slices=n_slices, mode="train", seed=0)
    model.train()
    for mod in model.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0
    fd = make_feed_dict(batch, size, n_qry, n_slices, seed=seed)
    out = model({k: v.clone() for k, v in fd.items()})
    lp = F.l1_loss(out["sdf_pred"], fd["sdf"])
    li = F.l1_loss(out["slices_rec"], fd["img_slices"])
    lv = out["vgg_loss"]
    (lp + li + lv).backward()
    acc = ((out["sdf_pred"] >= 0) == (fd["sdf"] >= 0)).float().sum(dim=-1) / out["sdf_pred"].shape[1]
    rec = {"meta": np.array([batch, size, n_qry, n_slices, seed], dtype=np.int64),
           "losses": np.array([float(lp), float(li), float(lv), float(acc.mean())], dtype=np.float64),
           "sdf_pred": out["sdf_pred"].detach().numpy()}
    for k in ("img_input", "img_slices", "qry_norot", "sdf", "obj_rot_mat", "trans_mat_wo_rot_tp"):
        rec[k] = fd[k].numpy()
    rng = np.random.default_rng(7)
    names = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        g = p.grad.reshape(-1)
        idx = rng.integers(0, g.numel(), 32)
        rec["gn:" + k] = np.array([float(g.norm()), float(g.abs().max())])
        rec["gi:" + k] = idx
        rec["gv:" + k] = g[torch.from_numpy(idx)].numpy()
    rec["grad_names"] = np.array(names)
    for k, v in model.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            rec["bn:" + k] = v.numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("%-34s losses %s  %d grads  %.1f KB" % (name, rec["losses"], len(names), os.path.getsize(path) / 1024))


def grid_case():
    sys.path.insert(0, REG_SLICES)
    from src_convonet.common import make_3d_grid
    rec = {}
    for n in (2, 4, 5):
        rec["grid%d" % n] = make_3d_grid((-0.5,) * 3, (0.5,) * 3, (n,) * 3).numpy()
    np.savez_compressed(os.path.join(OUT, "make_3d_grid.npz"), **rec)
    print("make_3d_grid.npz")


if __name__ == "__main__":
    # G1: BASELINE config 1 shape (1 sample, 64^2, 4 slices, 1k queries), train-mode rotation path
    run_case("g1_c1_s64_n4_q1000_train", 1, 64, 1000, 4, "train", seed=11)
    # G2/G3: 12 slices, mode='test' (y/z flip, no rotation) + stage tensors
    run_case("g2_s64_n12_q2048_test", 1, 64, 2048, 12, "test", seed=12, with_stages=True,
             store_slices=False)
    # batch-major flattening check (B=2), non-square-free small size
    run_case("g3_s32_n12_q512_b2_train", 2, 32, 512, 12, "train", seed=13, with_stages=True)
    # 128^2 (the reference's released resolution), few queries
    run_case("g4_s128_n12_q256_test", 1, 128, 256, 12, "test", seed=14, store_slices=False)
    grid_case()
    train_case("g5_train_s32_n12_q128_b2", 2, 32, 128, 12, seed=15)
