"""Generates tests/golden/gt*_train_*.npz: one TRAIN-mode forward/backward of the REAL reference Slices3DGTModel
(reg_slices/src/model_gt.py imported through oracle/ref_import.py) with the loss / accuracy of
reg_slices/train_gt.py:21-36.  Dropout is forced to 0 so the run is deterministic; BatchNorm uses batch statistics.
Run from the repo root:

    python tests/golden/make_golden_gt_train.py

Fixtures are data only: inputs, loss, accuracy, sdf_pred, per-tensor gradient norms plus 32 sampled entries, and
the updated BatchNorm running statistics.  Weights are regenerated from slice3d_amd.weights.seeded_array.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from make_golden_gt import _forward  # noqa: E402
from oracle.ref_import import build_reference_gt_model  # noqa: E402
from slice3d_amd.synth import make_feed_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def train_case(name, batch, size, n_qry, n_slices, seed):
    model = build_reference_gt_model(n_slices=n_slices, mode="train", img_size=size, seed=0)
    model.train()
    for mod in model.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0
    fd = make_feed_dict(batch, size, n_qry, n_slices, seed=seed)
    # size 128: the reference forward as is (conv_last + classifier run, feat_global dropped);
    # other sizes: the same forward with the unused feat_global branch skipped (the classifier needs 128^2)
    sdf = _forward(model, {k: v.clone() for k, v in fd.items()}, size)
    loss = F.l1_loss(sdf, fd["sdf"])
    loss.backward()
    acc = ((sdf >= 0) == (fd["sdf"] >= 0)).float().sum(dim=-1) / sdf.shape[1]
    rec = {"meta": np.array([batch, size, n_qry, n_slices, seed], dtype=np.int64),
           "losses": np.array([float(loss), float(acc.mean())], dtype=np.float64),
           "sdf_pred": sdf.detach().numpy()}
    for k in ("img_slices", "qry_norot", "sdf", "obj_rot_mat", "trans_mat_wo_rot_tp"):
        rec[k] = fd[k].numpy()
    rng = np.random.default_rng(9)
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


if __name__ == "__main__":
    train_case("gt3_train_s128_n12_q96_b1", 1, 128, 96, 12, seed=41)
    train_case("gt4_train_s32_n12_q130_b2", 2, 32, 130, 12, seed=42)
