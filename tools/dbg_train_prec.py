"""Per-parameter gradient difference between the f32 and f16x3 training paths (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
q = int(sys.argv[1]) if len(sys.argv) > 1 else 200
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
fd = {k: v.cuda() for k, v in make_feed_dict(1, size, q, 12, seed=77).items()}
res = {}
for prec in ("f32", "f16x3", "f32b"):
    m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
    tr = HipTrainer(m, prec=prec[:5].replace("f32b", "f32"))
    losses = tr.forward_backward(fd).cpu().numpy().copy()
    res[prec] = (losses, tr.grad_flat.cpu().clone(), tr)
    print(prec, losses)
tr = res["f32"][2]
for k, p in zip(tr.names, tr.params):
    off, n = tr.offsets[k], p.numel()
    a, b, c = (res[x][1][off:off + n] for x in ("f32", "f16x3", "f32b"))
    print("%-50s |g| %.3e  f16x3 rel %.2e  f32-rerun rel %.2e" % (k, a.norm(), (a - b).norm() / (a.norm() + 1e-30), (a - c).norm() / (a.norm() + 1e-30)))
