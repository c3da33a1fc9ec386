"""Loss trajectories of the f32 and f16x3 training paths on one fixed batch (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
S, Q, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fd = make_feed_dict(1, S, Q, 12, seed=1, device="cuda")
for prec in ("f32", "f16x3"):
    m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
    tr = HipTrainer(m, prec=prec)
    tr.dropout = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    out = []
    for i in range(n):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.time()
        out.append(tr.train_step(fd))
    torch.cuda.synchronize()
    print(prec, "ms/step %.1f" % ((time.time() - t0) / (n - 2) * 1e3), " ".join("%.4f" % (o[0]) for o in out))
