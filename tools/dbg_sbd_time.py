"""Timing experiments on sample_bwd_dense_kernel: S3D_SBD_DBG bits switch phases off (results are then wrong)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
fd = make_feed_dict(4, 256, 100000, 12, seed=1, device="cuda")
m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
tr = HipTrainer(m, prec="f16x3", dropout=0.1, seed=3)
for _ in range(2):
    tr.forward_backward(fd)
for dbg in sys.argv[1:] or ["0", "1", "2", "4", "8", "15", "off"]:
    if dbg == "off":
        os.environ["S3D_SBD_OFF"] = "1"
    else:
        os.environ["S3D_SBD_OFF"] = "0"; os.environ["S3D_SBD_DBG"] = dbg
    tr.forward_backward(fd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        tr.forward_backward(fd)
    torch.cuda.synchronize()
    print("S3D_SBD_DBG=%s: %.2f ms per step" % (dbg, (time.perf_counter() - t0) / 4 * 1e3))
