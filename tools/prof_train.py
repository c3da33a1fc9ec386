import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
tr = HipTrainer(m, prec=os.environ.get('S3D_PREC', 'f16x3'))
tr.dropout = float(os.environ.get('S3D_DROPOUT', '0.1'))
fd = make_feed_dict(int(os.environ.get("S3D_B", "4")), 256, 100000, 12, seed=1, device="cuda")
for _ in range(3):
    print(tr.train_step(fd))
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.train_step(fd)
torch.cuda.synchronize()
print("train step wall time: %.1f ms (5 steps)" % ((time.perf_counter() - t0) / 5 * 1e3))
