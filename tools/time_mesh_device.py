"""reconstruct.py's per-object work at its default options (256^2 image, MISE 64 -> 256) on the device mesh path vs the
host mesh path: seconds per mesh and where they go (tools/time_mesh_device.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slice3d_amd.generator import Generator3D
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded
m = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
fd = make_feed_dict(1, 256, 16, 12, seed=3, with_slices=False, device="cuda")
for backend in ("device", "host"):
    for res0, ups in ((64, 2), (256, 0)):
        g = Generator3D(m, threshold=0.5, resolution0=res0, upsampling_steps=ups, pred_type="sdf", mesh_backend=backend)
        g.generate_mesh(fd)
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            mesh, st = g.generate_mesh(fd)
            torch.cuda.synchronize(); ts.append(time.time() - t0)
        print("%-6s res0=%3d up=%d: %.3f s/mesh (best of 3)  eval %.3f s, marching cubes %.3f s, %d verts %d faces %s"
              % (backend, res0, ups, min(ts), st["time (eval points)"], st["time (marching cubes)"], len(mesh.vertices),
                 len(mesh.faces), {k: v for k, v in st.items() if k.startswith("mise")}))
