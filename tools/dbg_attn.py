import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
m16 = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
m32 = load_seeded(Slices3DRegModel(n_slices=12, mode="test"), 0).cuda().eval()
for q in (16, 32, 3000):
    fd = {k: v.cuda() for k, v in make_feed_dict(1, 64, q, 12, seed=55, with_slices=False).items()}
    a = m16(fd)["sdf_pred"]; a2 = m16(fd)["sdf_pred"]; b = m32(fd)["sdf_pred"]
    d = (a - b).abs()
    print(q, "max err %.3e  rerun diff %.3e  frac bad %.3f" % (float(d.max()), float((a - a2).abs().max()), float((d > 1e-4).float().mean())), "first bad idx", (d[0] > 1e-4).nonzero()[:8].flatten().tolist())
