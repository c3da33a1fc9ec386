"""Golden outputs of the REAL reference latent-diffusion UNetModel (gen_slices/ldm/.../openaimodel.py) with
name-seeded weights, for a small configuration and for the Slice3D configuration
(objaverse-ldm-kl-8.yaml:22-34).  Inputs are regenerated from a seed by tests/helpers.ldm_inputs on both sides;
only the reference's output is stored.  Authoring container only:

    python tests/golden/make_golden_ldm.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import ldm_inputs  # noqa: E402
from oracle.ref_import import LDM_FULL, LDM_SMALL, build_reference_ldm_unet  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(name, cfg, batch, seed):
    model = build_reference_ldm_unet(cfg)
    x, t, cf = ldm_inputs(cfg, batch, seed)
    cf = dict(cf)
    for k in ("f1", "f2", "f3", "f4", "f5"):   # the reference indexes all five; absent blocks never read theirs
        cf.setdefault(k, torch.zeros(1))
    with torch.no_grad():
        y = model(x, t, c_fmaps=cf)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), y=y.numpy(), meta=np.array([batch, seed]))
    print(name, tuple(y.shape), "mean |y| %.4f" % float(y.abs().mean()))
    return model


if __name__ == "__main__":
    m = run("ldm_small_b2", LDM_SMALL, 2, 5)
    json.dump({k: list(v.shape) for k, v in m.state_dict().items()},
              open(os.path.join(OUT, "state_dict_keys_ldm_small.json"), "w"), indent=0)
    m = run("ldm_full_b1", LDM_FULL, 1, 6)
    json.dump({k: list(v.shape) for k, v in m.state_dict().items()},
              open(os.path.join(OUT, "state_dict_keys_ldm_full.json"), "w"), indent=0)
