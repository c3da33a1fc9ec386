"""Shader clock and socket power while the LDM denoise step is replayed (HIP graph) for a few seconds — is the step's chain of
~10 us kernels running at the clock the long decoder kernels get?  Usage: python tools/ldm_clock.py [batch] [latent]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import ldm_inputs
from test_ldm import LDM_FULL
from slice3d_amd.ldm_unet import UNetModel
from slice3d_amd.weights import load_seeded
from bench import _ClockSampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = dict(LDM_FULL, image_size=SIZE)
m = load_seeded(UNetModel(**cfg), 0).cuda().eval()
x, t, cf = ldm_inputs(cfg, B, 1)
x, t, cf = x.cuda(), t.cuda(), {k: v.cuda() for k, v in cf.items()}
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        y = m(x, t, c_fmaps=cf)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    yg = m(x, t, c_fmaps=cf)
g.replay(); torch.cuda.synchronize()
n = 800
with _ClockSampler() as cs:
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
mhz, watts = cs.result()
print("LDM step B=%d %dx%d: %.3f ms/step over %d replays; sclk %.0f MHz, %.0f W" % (B, SIZE, SIZE, ms, n, mhz or -1, watts or -1))
