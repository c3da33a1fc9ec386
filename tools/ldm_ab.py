"""LDM denoise step (batch 1, 64x64 latent, DDIM loop with HIP-graph replay as bench.py times it) with the round-5 switches:
fuse_gn (GroupNorm inside the 3x3 convolution's staging path) x branch_streams (skip convolutions as a parallel graph branch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from slice3d_amd.ldm_sampler import DDIMSampler
from slice3d_amd.ldm_unet import UNetModel
from slice3d_amd.weights import load_seeded

cfg = dict(image_size=64, in_channels=8, out_channels=4, model_channels=192, attention_resolutions=[1, 2, 4, 8],
           num_res_blocks=2, channel_mult=[1, 2, 2, 4, 4], num_heads=8, use_scale_shift_norm=True, resblock_updown=True)
g = torch.Generator().manual_seed(0)
lx = torch.randn(1, 8, 64, 64, generator=g).cuda()
lc = {k: (torch.randn(1, c, r, r, generator=g) * 0.5).cuda()
      for k, (c, r) in (("f1", (192, 64)), ("f2", (384, 32)), ("f3", (384, 16)), ("f4", (768, 8)), ("f5", (768, 4)))}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for rep in range(2):
    for prec in ("f16x3", "f16"):
        for fuse, branch, defer in ((False, False, False), (True, False, False), (True, False, True), (True, True, True)):
            if True:
                um = load_seeded(UNetModel(prec=prec, fuse_gn=fuse, branch_streams=branch, defer_finish=defer, **cfg), 0).cuda().eval()
                smp = DDIMSampler(um)
                x_T, cc = lx[:, :4].contiguous(), lx[:, 4:].contiguous()
                gen = torch.Generator(device="cuda").manual_seed(0)
                smp.sample(200, x_T, cc, lc, eta=1.0, generator=gen, n_steps=3)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                smp.sample(200, x_T, cc, lc, eta=1.0, generator=gen, n_steps=steps)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / steps * 1e3
                print("prec %-5s fuse_gn %-5s defer_finish %-5s branch %-5s graph %-5s  %.3f ms/step" % (prec, fuse, defer, branch, bool(smp._graph), ms), flush=True)
                del um, smp
