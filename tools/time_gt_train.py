"""Slices3DGTModel training step time at the reference's default options (reg_slices/options.py: img_size 128,
n_qry 256, n_slices 12, n_bs 16) and at a query-heavy shape.  S3D_GT_STEPS=n limits the run for profiling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd.models_gt import Slices3DGTModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.trainer import HipGtTrainer
from slice3d_amd.weights import load_seeded
steps = int(os.environ.get("S3D_GT_STEPS", "6"))
cases = [(16, 128, 256)] if "S3D_GT_STEPS" in os.environ else [(16, 128, 256), (4, 128, 4096), (1, 256, 100000)]
for B, S, Q in cases:
    fd = make_feed_dict(B, S, Q, 12, seed=1, device="cuda")
    m = load_seeded(Slices3DGTModel(img_size=S, n_slices=12, mode="train"), 0).cuda()
    tr = HipGtTrainer(m, prec=os.environ.get("S3D_PREC", "f16x3"), dropout=0.1)
    for i in range(2 + steps):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.time()
        out = tr.train_step(fd)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    print("B=%d S=%d Q=%d: %.1f ms/step, %.1f samples/s  (loss %.4f acc %.3f, workspace %.1f GB)"
          % (B, S, Q, ms, B / ms * 1e3, out[0], out[1], tr._ws.numel() / 2**30))
    del tr, m, fd
    torch.cuda.empty_cache()
