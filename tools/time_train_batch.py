"""Training step time vs objects per step (BASELINE C2 allows B = 1..4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
for B in (1, 4):
    fd = make_feed_dict(B, 256, 100000, 12, seed=1, device="cuda")
    m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
    tr = HipTrainer(m, prec="f16x3", dropout=0.1)
    for i in range(5):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.time()
        tr.train_step(fd)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 3 * 1e3
    print("B=%d: %.1f ms/step, %.2f samples/s" % (B, ms, B / ms * 1e3))
    del tr, m, fd
    torch.cuda.empty_cache()
    # inference throughput with B objects per step
    mi = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
    fdi = make_feed_dict(B, 256, 100000, 12, seed=2, with_slices=False, device="cuda")
    for i in range(6):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.time()
        code = mi.encode(fdi); out = mi.decode_sdf(fdi["qry_norot"], code)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 4 * 1e3
    print("   inference B=%d: %.2f ms/step, %.2f M q/s" % (B, ms, B * 0.1 / ms * 1e3))
    del mi, fdi
    torch.cuda.empty_cache()
