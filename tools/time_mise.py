"""Where the MISE-driven evaluation of Generator3D.generate_value_grid spends its time (host MISE vs device decode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slice3d_amd.generator import Generator3D
from slice3d_amd.mesh import MISE
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded
m = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
fd = make_feed_dict(1, 256, 16, 12, seed=3, with_slices=False, device="cuda")
g = Generator3D(m, threshold=0.5, resolution0=64, upsampling_steps=2, pred_type="sdf")
for rep in range(2):
    tq = te = tu = th = 0.0
    npts = 0
    torch.cuda.synchronize(); t_all = time.time()
    code = g.encode(fd)
    mise = MISE(64, 2, 0.0)
    t0 = time.time(); points = mise.query(); tq += time.time() - t0
    rounds = 0
    while points.shape[0] != 0:
        rounds += 1; npts += points.shape[0]
        t0 = time.time()
        pf = 1.1 * (points.astype(np.float32) / mise.resolution - 0.5)
        d = dict(fd); d["qry_norot"] = torch.from_numpy(pf).unsqueeze(0).cuda()
        th += time.time() - t0
        t0 = time.time()
        g.chunk_size = 1 << 18
        v = g.eval_points(d, code); torch.cuda.synchronize()
        te += time.time() - t0
        t0 = time.time(); vals = v.cpu().numpy().astype(np.float64); th += time.time() - t0
        t0 = time.time(); mise.update(points, vals); tu += time.time() - t0
        t0 = time.time(); points = mise.query(); tq += time.time() - t0
    t0 = time.time(); grid = mise.to_dense(); td = time.time() - t0
    print("rounds %d points %d: total %.3f s | query %.3f update %.3f to_dense %.3f | host<->device %.3f | decode %.3f (%.2f M q/s)"
          % (rounds, npts, time.time() - t_all, tq, tu, td, th, te, npts / te / 1e6))
