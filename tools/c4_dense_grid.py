"""BASELINE config 4: dense 256^3 grid evaluation (16.7 M queries/object) + marching cubes, 1 MI355X."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.generator import Generator3D
for prec in ("f16x3", "f32"):
    m = load_seeded(Slices3DRegModel(img_size=256, n_slices=12, mode="test", prec=prec), 0).cuda().eval()
    fd = {k: v.cuda() for k, v in make_feed_dict(1, 256, 16, 12, seed=7, with_slices=False).items()}
    gen = Generator3D(m, resolution0=256, upsampling_steps=0, pred_type="sdf")
    gen.generate_value_grid(fd); torch.cuda.synchronize()
    t0 = time.time(); grid = gen.generate_value_grid(fd); torch.cuda.synchronize(); t1 = time.time()
    print("%s: dense 256^3 (16.78 M queries) encode+decode+D2H %.3f s -> %.2f M q/s" % (prec, t1 - t0, 256**3 / (t1 - t0) / 1e6))
st = {}
mesh = gen.extract_mesh(grid, stats_dict=st)
print("marching cubes (host C++):", st, len(mesh.vertices), "verts", len(mesh.faces), "faces")
gen2 = Generator3D(m, resolution0=64, upsampling_steps=2, pred_type="sdf")
t0 = time.time(); g2 = gen2.generate_value_grid(fd); torch.cuda.synchronize(); print("MISE 64->256 value grid %.3f s" % (time.time() - t0), g2.shape)
