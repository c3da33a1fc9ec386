"""Slices3DGTModel inference: encode the 12 given slice images (VGG16-BN pyramid + folded maps) + decode 100 k queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd.models_gt import Slices3DGTModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded
for B, S, Q in ((1, 128, 100000), (4, 128, 100000), (1, 256, 100000)):
    m = load_seeded(Slices3DGTModel(img_size=S, n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
    fd = make_feed_dict(B, S, Q, 12, seed=2, device="cuda")
    for i in range(7):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.time()
        code = m.encode(fd)
        out = m.decode_sdf(fd["qry_norot"], code, trans_mat_wo_rot_tp=fd["trans_mat_wo_rot_tp"])
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 5 * 1e3
    print("B=%d S=%d Q=%d: %.2f ms/step, %.2f M query-points/s" % (B, S, Q, ms, B * Q / ms / 1e3))
    del m, fd
    torch.cuda.empty_cache()
