"""Golden output of the REAL reference latent-diffusion UNetModel at the 128x128 latent size BASELINE configs[4] names
("256^2 slice generation": kl-f8 autoencoder of a 1024^2 4x3 mosaic of 256^2 slices -> 128x128x4 latent; 16 384-token
attention at full resolution), objaverse-ldm-kl-8.yaml:22-34 otherwise.  Name-seeded weights, seeded inputs
(tests/helpers.ldm_inputs); only the reference's output is stored.  Authoring container only (needs ~30 GB of RAM for
the reference's materialised 16 384 x 16 384 x 8 attention weights):

    python tests/golden/make_golden_ldm128.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import ldm_inputs  # noqa: E402
from oracle.ref_import import LDM_FULL, build_reference_ldm_unet  # noqa: E402

if __name__ == "__main__":
    cfg = dict(LDM_FULL, image_size=128)
    model = build_reference_ldm_unet(cfg)
    x, t, cf = ldm_inputs(cfg, 1, 7)
    with torch.no_grad():
        y = model(x, t, c_fmaps=dict(cf))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ldm_full128_b1.npz")
    np.savez_compressed(out, y=y.numpy(), meta=np.array([1, 7]))
    print(out, tuple(y.shape), "mean |y| %.4f" % float(y.abs().mean()))
