"""Golden vectors for the data-parallel exchange step ("parity oracle"): the REAL reference, train
mode, dropout pinned to 0, run on each one-sample SHARD of a two-sample batch exactly as a data-parallel rank would
(its own batch statistics in BatchNorm, PyTorch DDP's default), giving per-shard gradients g_0, g_1; the all-reduced
gradient must be (g_0 + g_1)/2, and torch.optim.Adam (train.py:136 settings) stepped with it gives the parameters
every rank must hold afterwards.  Stored in full for a set of SMALL tensors of every part of the model (norms, biases,
fc_out / fc_p, slice embeddings, BatchNorm affine: 27 k floats), plus norm + 32 sampled entries for all tensors.

    python tests/golden/make_golden_ddp.py        (authoring container: imports /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ref_import import build_reference_model  # noqa: E402
from slice3d_amd.synth import make_feed_dict  # noqa: E402

FULL = ("fc_out.0.weight", "fc_out.0.bias", "fc_p.weight", "fc_p.bias", "fc_s.bias",
        "att_decoder.layers.0.norm1.weight", "att_decoder.layers.1.norm2.bias", "att_decoder.layers.2.linear2.bias",
        "att_decoder.layers.0.self_attn.in_proj_bias", "att_decoder.layers.2.self_attn.out_proj.bias",
        "slices_generator.emds.weight", "slices_generator.outc.conv.weight", "slices_generator.outc.conv.bias",
        "slices_generator.up4.conv.double_conv.1.weight", "slices_generator.up1.conv.double_conv.4.bias",
        "slices_generator.trans_up4.bias", "slices_generator.up2.up.bias", "slices_generator.trans_c.bias",
        "slices_generator.down1.1.weight", "slices_generator.down3.18.bias", "slices_generator.down5.38.weight",
        "slices_generator.down4.24.bias", "slices_generator.down1.0.weight")


def shard_grads(fd, n_slices):
    model = build_reference_model(n_slices=n_slices, mode="train", seed=0)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    x = model({k: v.clone() for k, v in fd.items()})
    loss = (torch.nn.functional.l1_loss(x["sdf_pred"], fd["sdf"]) +
            torch.nn.functional.l1_loss(x["slices_rec"], fd["img_slices"]) + x["vgg_loss"])      # train.py:29-47
    loss.backward()
    return model, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def main():
    b, s, q, ns, seed = 2, 32, 160, 12, 21
    fd = make_feed_dict(b, s, q, ns, seed=seed)
    out = {"meta": np.array([b, s, q, ns, seed]), "full_names": np.array(FULL)}
    for k, v in fd.items():
        out[k] = v.numpy()
    grads = []
    for r in range(b):
        model, g = shard_grads({k: v[r:r + 1].contiguous() for k, v in fd.items()}, ns)
        grads.append(g)
    names = sorted(grads[0])
    out["grad_names"] = np.array(names)
    rng = np.random.default_rng(3)
    for k in names:
        mean = (grads[0][k] + grads[1][k]) / 2
        flat = mean.reshape(-1).numpy()
        idx = rng.choice(flat.size, min(32, flat.size), replace=False)
        out["gn:" + k] = np.array([np.linalg.norm(flat), np.abs(flat).max()])
        out["gi:" + k], out["gv:" + k] = idx, flat[idx]
    # full tensors: per-shard gradients, the parameter before and after one Adam step with the mean gradient
    params = dict(model.named_parameters())
    init = {k: params[k].detach().clone() for k in FULL}
    leaves = [init[k].clone().requires_grad_(True) for k in FULL]
    opt = torch.optim.Adam(leaves, lr=3e-4)
    for leaf, k in zip(leaves, FULL):
        leaf.grad = (grads[0][k] + grads[1][k]) / 2
    opt.step()
    for leaf, k in zip(leaves, FULL):
        out["g0:" + k], out["g1:" + k] = grads[0][k].numpy(), grads[1][k].numpy()
        out["p0:" + k], out["p1:" + k] = init[k].numpy(), leaf.detach().numpy()
    path = os.path.join(HERE, "g6_ddp_shards_s32_n12_q160_b2.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", sum(init[k].numel() for k in FULL), "floats in full")


if __name__ == "__main__":
    main()
