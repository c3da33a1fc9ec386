"""Times one denoising step of the Slice3D latent-diffusion U-Net configuration (BASELINE configs[4])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import ldm_inputs
from test_ldm import LDM_FULL
from slice3d_amd.ldm_unet import UNetModel
from slice3d_amd.weights import load_seeded
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 64          # latent size: 64 (the bench's step) or 128 (256^2 slices)
LDM_FULL = dict(LDM_FULL, image_size=SIZE)
m = load_seeded(UNetModel(**LDM_FULL), 0).cuda().eval()
x, t, cf = ldm_inputs(LDM_FULL, B, 1)
x, t, cf = x.cuda(), t.cuda(), {k: v.cuda() for k, v in cf.items()}
for _ in range(3):
    y = m(x, t, c_fmaps=cf)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    y = m(x, t, c_fmaps=cf)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print("LDM denoise step B=%d, %dx%d latent: %.2f ms  (%.1f TFLOP/s algorithmic at 222 GFLOP/step/sample at 64x64; the convolutions "
      "scale with the pixel count, the attention with its square)" % (B, SIZE, SIZE, ms, 0.222 * B / ms * 1e3))

# the same step captured once into a HIP graph (static shapes, ~370 launches) and replayed
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
print("graph output matches eager:", float((yg - y).abs().max()))
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print("LDM denoise step B=%d, HIP-graph replay: %.2f ms" % (B, ms))
