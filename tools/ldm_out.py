"""Writes the output of one denoising step of the full gen_slices U-Net configuration on seeded inputs to a file (used by the
A/B tests that compare two builds / environment switches of the library bit for bit).  Usage: python tools/ldm_out.py OUT.pt [batch ...]
(one batch size: OUT.pt holds the tensor; several: a dict batch -> tensor from ONE model load)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import ldm_inputs
from test_ldm import LDM_FULL
from slice3d_amd.ldm_unet import UNetModel
from slice3d_amd.weights import load_seeded
BS = [int(a) for a in sys.argv[2:]] or [1]
m = load_seeded(UNetModel(**LDM_FULL), 0).cuda().eval()
outs = {}
for B in BS:
    x, t, cf = ldm_inputs(LDM_FULL, B, 1)
    y = m(x.cuda(), t.cuda(), c_fmaps={k: v.cuda() for k, v in cf.items()})
    torch.cuda.synchronize()
    outs[B] = y.cpu()
torch.save(outs[BS[0]] if len(BS) == 1 else outs, sys.argv[1])
