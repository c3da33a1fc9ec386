"""Stage times of Generator3D.generate_mesh (encode + MISE-driven decode + marching cubes) at the reference's
default reconstruct options (mc_res0 / mc_up_steps of options.py) and for the dense 256^3 grid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd.generator import Generator3D
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded
m = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
fd = make_feed_dict(1, 256, 16, 12, seed=3, with_slices=False, device="cuda")
for res0, up in ((int(os.environ.get("RES0", 64)), int(os.environ.get("UP", 2))), (256, 0)):
    g = Generator3D(m, threshold=0.5, resolution0=res0, upsampling_steps=up, pred_type="sdf")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        mesh, st = g.generate_mesh(fd)
        torch.cuda.synchronize(); t = time.time() - t0
    print("res0=%d up=%d: total %.3f s  eval %.3f  mc %.3f  verts %d faces %d" %
          (res0, up, t, st["time (eval points)"], st["time (marching cubes)"], len(mesh.vertices), len(mesh.faces)))
