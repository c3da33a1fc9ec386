"""Writes the output of one denoising step of the full gen_slices U-Net configuration on seeded inputs to a file (used by the
A/B tests that compare two builds / environment switches of the library bit for bit).  Usage: python tools/ldm_out.py OUT.pt [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import ldm_inputs
from test_ldm import LDM_FULL
from slice3d_amd.ldm_unet import UNetModel
from slice3d_amd.weights import load_seeded
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = load_seeded(UNetModel(**LDM_FULL), 0).cuda().eval()
x, t, cf = ldm_inputs(LDM_FULL, B, 1)
y = m(x.cuda(), t.cuda(), c_fmaps={k: v.cuda() for k, v in cf.items()})
torch.cuda.synchronize()
torch.save(y.cpu(), sys.argv[1])
