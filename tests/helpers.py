"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from slice3d_amd.weights import seeded_array

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

GOLDEN_CASES = ("g1_c1_s64_n4_q1000_train", "g2_s64_n12_q2048_test", "g3_s32_n12_q512_b2_train",
                "g4_s128_n12_q256_test")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    b, s, q, ns, seed = [int(v) for v in g["meta"]]
    g.update(batch=b, size=s, n_qry=q, n_slices=ns, seed=seed, mode=str(g["mode"]))
    return g


def golden_feed(g, device="cpu"):
    fd = {k: torch.from_numpy(g[k]).to(device) for k in
          ("img_input", "qry_norot", "obj_rot_mat", "trans_mat_wo_rot_tp")}
    if "img_slices" in g:
        fd["img_slices"] = torch.from_numpy(g["img_slices"]).to(device)
    return fd


def reference_state_dict_shapes(n_slices=12):
    """Key -> shape table of the reference Slices3DRegModel state_dict (244 tensors), derived from
    SURVEY.md 8(a)/8(b); test_state_dict.py checks the package's model reproduces it exactly."""
    from slice3d_amd.models import Slices3DRegModel
    m = Slices3DRegModel(n_slices=n_slices, backend="none")
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def seeded_sd_from_shapes(shapes, seed=0, dtype=torch.float32):
    sd = {}
    for k, shp in shapes.items():
        a = seeded_array(k, shp, seed)
        t = torch.from_numpy(a)
        sd[k] = t.to(dtype) if t.is_floating_point() else t
    return sd


def ldm_fmap_shapes(cfg):
    """(channels, side) of h after input blocks 0, 4, 7, 10, 12 of the LDM UNetModel (openaimodel.py:735-746)."""
    mc, side = cfg["model_channels"], cfg["image_size"]
    out, ch, idx = {"f1": (mc, side)}, mc, 0
    names = {4: "f2", 7: "f3", 10: "f4", 12: "f5"}
    for level, mult in enumerate(cfg["channel_mult"]):
        for _ in range(cfg["num_res_blocks"]):
            idx += 1
            ch = mult * mc
            if idx in names:
                out[names[idx]] = (ch, side)
        if level != len(cfg["channel_mult"]) - 1:
            idx += 1
            side //= 2
            if idx in names:
                out[names[idx]] = (ch, side)
    return out


def ldm_inputs(cfg, batch, seed):
    """Deterministic (CPU generator) inputs of one denoising step: x, timesteps, c_fmaps."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, cfg["in_channels"], cfg["image_size"], cfg["image_size"], generator=g)
    t = torch.randint(0, 1000, (batch,), generator=g)
    cf = {k: torch.randn(batch, c, s, s, generator=g) * 0.5 for k, (c, s) in ldm_fmap_shapes(cfg).items()}
    return x, t, cf


def check_grads_against_golden(z, grads, skip=(), tol=2e-2):
    """z: a *_train_* golden (gn:/gi:/gv: entries written by make_golden*.py from the real reference's .grad);
    grads: name -> flat numpy gradient.  Per tensor: the L2 norm within tol and the 32 sampled entries within
    tol * max|g|.  Tensors in `skip` (exactly-zero gradients, rounding noise on both sides) only need to be tiny.
    Returns the worst sampled error / max|g|."""
    worst = 0.0
    for k in z["grad_names"]:
        k = str(k)
        g = grads[k]
        norm, gmax = z["gn:" + k]
        if k in skip:
            assert np.abs(g).max() < 1e-4, k
            continue
        assert abs(np.linalg.norm(g) - norm) < tol * norm, (k, np.linalg.norm(g), norm)
        err = np.abs(g[z["gi:" + k]] - z["gv:" + k]).max() / gmax
        worst = max(worst, err)
        assert err < tol, (k, err)
    return worst
