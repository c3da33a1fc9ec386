"""Golden of the REAL reference DDIM sampler (gen_slices/ldm/models/diffusion/ddim.py) around the REAL reference
UNetModel with name-seeded weights: a complete 4-step sampling run (S = 4: timesteps 751, 501, 251, 1) (eta = 1, the reference's own loop, schedule and
update), the noise it drew, and the 200-step schedule arrays log_images_when_testing uses.  Authoring container only:

    python tests/golden/make_golden_ldm_ddim.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import ldm_inputs  # noqa: E402
from oracle.ref_import import LDM_SMALL, build_reference_ldm_unet  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    cfg = LDM_SMALL
    unet = build_reference_ldm_unet(cfg)
    import ldm.models.diffusion.ddim as ddim_mod            # the reference module (sys.path set by build_reference_ldm_unet)
    from ldm.modules.diffusionmodules.util import make_beta_schedule

    x8, _, cf = ldm_inputs(cfg, 2, 11)
    x_T, c_concat = x8[:, :4].contiguous(), x8[:, 4:].contiguous()
    cf = dict(cf)
    for k in ("f1", "f2", "f3", "f4", "f5"):
        cf.setdefault(k, torch.zeros(1))

    class Model:      # what DDIMSampler needs of LatentDiffusion (ddpm.py:118-160, 995-1004, 1461-1466)
        num_timesteps = 1000
        device = torch.device("cpu")
        _b = np.asarray(make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0155))   # float64 (util.py:21-25)
        betas = torch.tensor(_b, dtype=torch.float32)                                               # ddpm.py:135-139
        alphas_cumprod = torch.tensor(np.cumprod(1.0 - _b, axis=0), dtype=torch.float32)
        alphas_cumprod_prev = torch.tensor(np.append(1.0, np.cumprod(1.0 - _b, axis=0)[:-1]), dtype=torch.float32)

        def apply_model(self, x, t, c):
            return unet(torch.cat([x] + c["c_concat"], dim=1), t, c_fmaps=c["c_fmaps"])

    class CpuSampler(ddim_mod.DDIMSampler):     # the reference moves every buffer to "cuda": keep them where they are
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    drawn = []
    real_noise_like = ddim_mod.noise_like

    def recording_noise_like(shape, device, repeat=False):
        n = real_noise_like(shape, device, repeat)
        drawn.append(n.clone())
        return n
    ddim_mod.noise_like = recording_noise_like
    torch.manual_seed(1234)
    s = CpuSampler(Model())
    with torch.no_grad():
        samples, inter = s.sample(4, 2, (4, cfg["image_size"], cfg["image_size"]), conditioning={"c_concat": [c_concat], "c_fmaps": cf},
                                  eta=1.0, x_T=x_T, verbose=False, log_every_t=1)
    s200 = CpuSampler(Model())
    s200.make_schedule(200, ddim_eta=1.0, verbose=False)
    np.savez_compressed(os.path.join(OUT, "ldm_ddim_small_b2.npz"), samples=samples.numpy(),
                        x_inter=np.stack([t.numpy() for t in inter["x_inter"]]),
                        pred_x0=np.stack([t.numpy() for t in inter["pred_x0"]]),
                        noises=np.stack([t.numpy() for t in drawn]), timesteps4=np.asarray(s.ddim_timesteps),
                        timesteps200=np.asarray(s200.ddim_timesteps), sigmas200=np.asarray(s200.ddim_sigmas),
                        alphas200=np.asarray(s200.ddim_alphas), alphas_prev200=np.asarray(s200.ddim_alphas_prev),
                        meta=np.array([2, 11]))
    print("samples", tuple(samples.shape), "x_inter", len(inter["x_inter"]), "noises", len(drawn), "mean |x0| %.4f" % float(samples.abs().mean()))
